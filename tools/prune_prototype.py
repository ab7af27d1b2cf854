#!/usr/bin/env python3
"""CPU study (numpy, no GPU) of one pass-1 idea: skip the score-row gather for embeddings whose centroid scores below a
threshold theta against every query token, bounding their contribution by theta_up = (theta + max|Q.r|) * max inv_norm.
Prints, per query and theta, the share of candidate embeddings still gathered and the number of passages the two-pass
selection would then keep.  Result on the synthetic corpus: not viable (profiles/r03_experiments.md).
usage: python tools/prune_prototype.py [n_passages]"""
import sys, time, numpy as np
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from colbert_jl_amd import synthetic
t0 = time.time()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
idx = synthetic.make_index(seed=2024, n_docs=N, n_blocks=8)
C = np.asarray(idx["centroids"], dtype=np.float32)            # (dim, K)
K = C.shape[1]
codes = np.asarray(idx["codes"]).astype(np.int64) - 1
doclens = np.asarray(idx["doclens"]).astype(np.int64)
off = np.concatenate([[0], np.cumsum(doclens)])
res = np.asarray(idx["residuals"])                             # (32, n_emb) uint8
w = np.asarray(idx["bucket_weights"], dtype=np.float32)
ivf = np.asarray(idx["ivf"]).astype(np.int64) - 1
ivfl = np.asarray(idx["ivf_lengths"]).astype(np.int64)
ivfo = np.concatenate([[0], np.cumsum(ivfl)])
emb2pid = np.repeat(np.arange(N), doclens)
print("index", N, K, codes.size, "t", round(time.time() - t0, 1))
Q = synthetic.make_topic_queries(C, seed=77, n_queries=8, T=32)   # (dim, T, nq)
k = 1000 if N >= 100000 else 100
def unpack(resb):     # (32, n) uint8 -> (128, n) idx
    out = np.empty((128, resb.shape[1]), dtype=np.uint8)
    for j in range(32):
        b = resb[j]
        for i in range(4):
            out[4 * j + i] = (b >> (2 * i)) & 3
    return out
for qi in range(4):
    q = Q[:, :, qi]                      # (dim, T)
    S = q.T @ C                          # (T, K)
    top2 = np.argsort(-S, axis=1)[:, :2]
    cids = np.unique(top2)
    eids = np.unique(np.concatenate([ivf[ivfo[c]:ivfo[c + 1]] for c in cids]))
    pids = np.unique(emb2pid[eids])
    # all embeddings of the candidate passages
    e_all = np.concatenate([np.arange(off[p], off[p + 1]) for p in pids])
    seg = np.repeat(np.arange(pids.size), doclens[pids])
    cc = codes[e_all]
    r = w[unpack(res[:, e_all])]          # (128, n)
    D = C[:, cc] + r
    inv = 1.0 / (np.sqrt((D * D).sum(0)) + np.float32(1.19e-7))
    A = (S[:, cc] + q.T @ r) * inv        # (T, n) "approx" = exact here
    starts = np.concatenate([[0], np.cumsum(doclens[pids])[:-1]])
    full_t = np.maximum.reduceat(A, starts, axis=1)          # (T, npass)
    full = full_t.sum(0)
    tau = np.sort(full)[-k] if full.size >= k else -np.inf
    cmax = S.max(0)                        # per-centroid max over tokens
    qr = np.abs(q.T @ r).max()
    print(f"q{qi}: cand passages {pids.size} emb {e_all.size} tau {tau:.2f}  max|Q.r| {qr:.3f} inv max {inv.max():.2f}")
    for theta in (0.15, 0.2, 0.25, 0.3, 0.35, 0.4):
        keep_c = cmax >= theta
        keep = keep_c[cc]
        Am = np.where(keep[None, :], A, -np.inf)
        L_t = np.maximum.reduceat(Am, starts, axis=1)
        theta_up = (theta + qr) * inv.max()
        U_t = np.maximum(L_t, theta_up)
        L = np.where(np.isfinite(L_t), L_t, -1e9).sum(0)
        U = U_t.sum(0)
        tauL = np.sort(L)[-k] if L.size >= k else -np.inf
        sel = int((U >= tauL - 0.09).sum())
        base = int((full >= tau - 0.09).sum())
        # passages where some token's window reaches the pruned bound -> all rows
        allrows = int(((L_t - 0.003) <= theta_up).any(0)[(U >= tauL - 0.09)].sum())
        print(f"   theta {theta:.2f}: centroids kept {keep_c.mean()*100:5.1f}%  cand emb gathered {keep.mean()*100:5.1f}%  selected {sel} (unpruned {base})  tauL-tau {tauL - tau:+.3f}  selected-with-all-rows {allrows}")
