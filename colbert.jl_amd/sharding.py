"""Passage sharding for multi-GPU search: contiguous pid ranges balanced by embedding
count; each shard keeps the full centroid table and a local IVF.  Scores are per passage, so the union of
the shards' candidates and their scores equals the unsharded search; only the final top-k needs an
exchange (one all-gather of k (pid, score) records per rank)."""
from __future__ import annotations

import numpy as np

from . import synthetic


def shard_bounds(doclens: np.ndarray, nranks: int) -> np.ndarray:
    """Passage boundaries [nranks+1] splitting the collection into contiguous ranges of ~equal embeddings."""
    cum = np.concatenate([[0], np.cumsum(np.asarray(doclens, dtype=np.int64))])
    targets = cum[-1] * np.arange(1, nranks) / nranks
    cuts = np.searchsorted(cum, targets, side="left")
    return np.concatenate([[0], cuts, [len(doclens)]]).astype(np.int64)


def shard_index(index: dict, rank: int, nranks: int):
    """-> (sub-index with the Searcher's fields, pid_offset = passages before this shard)."""
    b = shard_bounds(index["doclens"], nranks)
    lo, hi = int(b[rank]), int(b[rank + 1])
    cum = np.concatenate([[0], np.cumsum(index["doclens"])])
    e_lo, e_hi = int(cum[lo]), int(cum[hi])
    codes = index["codes"][e_lo:e_hi]
    K = index["centroids"].shape[1]
    ivf, ivf_lengths = synthetic.build_ivf(codes, K)
    sub = dict(index, doclens=index["doclens"][lo:hi], codes=codes,
               residuals=np.asfortranarray(index["residuals"][:, e_lo:e_hi]), ivf=ivf, ivf_lengths=ivf_lengths)
    return sub, lo


def merge_topk_host(pid_lists, score_lists, k: int):
    """Host restatement of the cross-shard merge (score desc, pid asc); pads (pid 0, -inf) are dropped."""
    p = np.concatenate([np.asarray(x, dtype=np.int64) for x in pid_lists])
    s = np.concatenate([np.asarray(x, dtype=np.float32) for x in score_lists])
    keep = p > 0
    p, s = p[keep], s[keep]
    order = np.lexsort((p, -s.astype(np.float64)))[:k]
    return p[order], s[order]
