#!/usr/bin/env python3
"""Round 6: how evenly pass 1 deals a batch to its waves on the headline corpus (CPU, ~3 min): steps per wave under the contiguous
split by passages, under a split by steps, and with the queries dealt to the XCD groups by size (profiles/r06_wave_balance.txt)."""
import sys, time, numpy as np, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import colbert_jl_amd as clb
from colbert_jl_amd import synthetic
t0=time.time()
K = synthetic.num_partitions_for(1_000_000, 80.0)
idx = synthetic.make_index(seed=2024, n_docs=1_000_000, K=K, n_blocks=8, topical=True)
print('gen', time.time()-t0, flush=True)
Q = synthetic.make_topic_queries(idx["centroids"], seed=77, n_queries=32, T=32)
C = np.asarray(idx["centroids"])
doclens = np.asarray(idx["doclens"]).astype(np.int64)
emb2pid = np.repeat(np.arange(doclens.size), doclens)
ivf = np.asarray(idx["ivf"]).astype(np.int64) - 1
ivl = np.asarray(idx["ivf_lengths"]).astype(np.int64)
ivo = np.concatenate([[0], np.cumsum(ivl)])
NW = 384
per_query = []; per_query_bal = []; totals = []
steps_w = np.zeros((8, NW))      # [xcd group][wave in group] accumulated over its 4 queries
steps_bal = np.zeros((8, NW))
pend_w = np.zeros((8, NW))
for b in range(32):
    S = Q[:, :, b].T @ C
    top = np.argpartition(-S, 2, axis=1)[:, :2].ravel()
    cids = np.unique(top)
    eids = np.concatenate([ivf[ivo[c]:ivo[c+1]] for c in cids])
    pids = np.unique(emb2pid[eids])
    L = doclens[pids]
    st = (L + 31) // 32
    n = pids.size
    per = (n + NW - 1) // NW
    cs = np.concatenate([[0], np.cumsum(st)])
    edges = np.minimum(n, np.arange(NW + 1) * per)
    w = cs[edges[1:]] - cs[edges[:-1]]
    steps_w[b % 8] += w
    per_query.append(w.astype(float)); totals.append(float(cs[-1]))
    pend_w[b % 8] += edges[1:] - edges[:-1]
    # balanced by steps: boundaries at equal step counts
    tgt = np.arange(NW + 1) * cs[-1] / NW
    e2 = np.searchsorted(cs, tgt, side='left'); e2[0] = 0; e2[-1] = n
    steps_bal[b % 8] += cs[e2[1:]] - cs[e2[:-1]]
    per_query_bal.append((cs[e2[1:]] - cs[e2[:-1]]).astype(float))
    if b < 3: print(b, n, int(L.sum()), 'steps', int(st.sum()), 'mean len', L.mean(), 'std', L.std(), 'wave steps mean/max', w.mean(), w.max(), flush=True)
for name, A in (('contiguous by passages', steps_w), ('balanced by steps', steps_bal)):
    wave = A.max() / A.mean()
    simd = A.reshape(8, 32, 3, 4).sum(2)      # wave id = 12*wg + k, SIMD = k % 4
    cu = A.reshape(8, 32, 12).sum(2)
    print(name, 'max/mean per wave %.3f per SIMD %.3f per CU %.3f' % (wave, simd.max() / simd.mean(), cu.max() / cu.mean()))

# ---- query -> XCD group assignment variants (per-wave contiguous split by passages kept)
import itertools
def run(assign, label, balanced=False):
    A = np.zeros((8, NW))
    for g in range(8):
        for b in assign[g]:
            A[g] += (per_query_bal[b] if balanced else per_query[b])
    cu = A.reshape(8, 32, 12).sum(2)
    print(label, 'per wave %.3f per CU %.3f group sums max/mean %.3f' % (A.max() / A.mean(), cu.max() / cu.mean(), A.sum(1).max() / A.sum(1).mean()))

nat = [[g, g + 8, g + 16, g + 24] for g in range(8)]
order = np.argsort(-np.array(totals))
snake = [[] for _ in range(8)]
for r, b in enumerate(order):
    rnd, pos = divmod(r, 8)
    g = pos if rnd % 2 == 0 else 7 - pos
    snake[g].append(int(b))
run(nat, 'natural           ')
run(snake, 'snake by steps    ')
run(nat, 'natural + balanced', True)
run(snake, 'snake + balanced  ', True)
