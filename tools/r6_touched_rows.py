#!/usr/bin/env python3
"""Round 6: how many rows of a query's centroid-score table its candidate embeddings touch on the headline corpus (CPU, ~2 min).
Answers why pass 1's row gather misses L2: the working set is the whole 8-MB table (profiles/r06_touched_rows.txt)."""
import sys, time, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import colbert_jl_amd as clb
from colbert_jl_amd import synthetic
t0=time.time()
K = synthetic.num_partitions_for(1_000_000, 80.0)
idx = synthetic.make_index(seed=2024, n_docs=1_000_000, K=K, n_blocks=8, topical=True)
print('gen', time.time()-t0, flush=True)
Q = synthetic.make_topic_queries(idx["centroids"], seed=77, n_queries=8, T=32)
C = np.asarray(idx["centroids"])  # (dim, K)
codes = np.asarray(idx["codes"]).astype(np.int64) - 1
doclens = np.asarray(idx["doclens"]).astype(np.int64)
off = np.concatenate([[0], np.cumsum(doclens)])
emb2pid = np.repeat(np.arange(doclens.size), doclens)
ivf = np.asarray(idx["ivf"]).astype(np.int64) - 1
ivl = np.asarray(idx["ivf_lengths"]).astype(np.int64)
ivo = np.concatenate([[0], np.cumsum(ivl)])
for b in range(4):
    S = Q[:, :, b].T @ C          # (32, K)
    top = np.argsort(-S, axis=1)[:, :2].ravel()
    cids = np.unique(top)
    eids = np.concatenate([ivf[ivo[c]:ivo[c+1]] for c in cids])
    pids = np.unique(emb2pid[eids])
    ce = np.concatenate([codes[off[p]:off[p+1]] for p in pids])
    u, cnt = np.unique(ce, return_counts=True)
    cs = np.sort(cnt)[::-1]
    cum = np.cumsum(cs) / cs.sum()
    print(f"query {b}: {pids.size} passages, {ce.size} embeddings, {u.size} distinct codes ({u.size*64/1e6:.2f} MB of fp16 rows); "
          f"rows covering 50/80/90/95/99 % of the accesses: {[int(np.searchsorted(cum, f)) + 1 for f in (0.5, 0.8, 0.9, 0.95, 0.99)]}", flush=True)
