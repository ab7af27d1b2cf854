"""Searcher / search -- the reference's src/searching.jl over the C ABI.

`Searcher(index_path)` loads an index directory, uploads it once into HBM (the reference keeps the
index in host memory, searching.jl:50-59) and `search` runs the whole post-encoder pipeline
(searching.jl:102-127) on the device in one library call."""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import numpy as np

from . import storage
from ._lib import BoundsError, ColBERTError, check, colmajor, fptr, i64, lib
from .config import ColBERTConfig


class Searcher:
    """struct Searcher (searching.jl:1-16).  Fields that the reference holds as arrays live in HBM
    behind `self._h`; the shapes/meta are kept for introspection."""

    def __init__(self, index_path: Optional[str] = None, *, config: Optional[ColBERTConfig] = None,
                 index: Optional[dict] = None, device: int = 0, pid_offset: int = 0, encoder=None):
        if index is None:
            if index_path is None:
                raise ValueError("Searcher needs an index_path or an in-memory index")
            config = ColBERTConfig.load(index_path)
            index = storage.load_index(index_path)
        self.config = config or ColBERTConfig(nbits=int(index["nbits"]), dim=int(index["dim"]))
        self.device = device
        self.encoder = encoder
        self.dim = int(index["dim"]); self.nbits = int(index["nbits"])
        if hasattr(index["codes"], "data_ptr"):
            self._create_from_device_arrays(index, pid_offset)
            return
        c = colmajor(index["centroids"], np.float32)
        w = np.ascontiguousarray(index["bucket_weights"], dtype=np.float32)
        dl = np.ascontiguousarray(index["doclens"], dtype=np.int64)
        co = np.ascontiguousarray(index["codes"], dtype=np.uint32)
        r = colmajor(index["residuals"], np.uint8)
        iv = np.ascontiguousarray(index["ivf"], dtype=np.int64)
        il = np.ascontiguousarray(index["ivf_lengths"], dtype=np.int64)
        if c.shape[0] != self.dim:
            raise ColBERTError("centroids must be (dim, K)")
        if r.shape != (self.dim // 8 * self.nbits, co.size):
            raise ColBERTError("residuals must be (dim/8*nbits, n_emb)")
        if il.size != c.shape[1]:
            raise ColBERTError("ivf_lengths must have one entry per centroid")
        self.num_centroids = c.shape[1]; self.num_docs = dl.size; self.num_embeddings = co.size
        self._h = C.c_void_p()
        check(lib().clb_searcher_create(device, i64(self.dim), C.c_int(self.nbits), i64(c.shape[1]), fptr(c), fptr(w),
                                        i64(dl.size), fptr(dl), i64(co.size), fptr(co), fptr(r), fptr(iv), fptr(il),
                                        i64(pid_offset), C.byref(self._h)))

    def _create_from_device_arrays(self, index: dict, pid_offset: int):
        """An index that was built in HBM (indexer.index_device): centroids (K, dim) float32, codes int32/uint32 [n],
        residuals uint8 (n, dim/8*nbits), ivf int64 [n] as CUDA tensors; doclens / ivf_lengths / bucket_weights on the
        host (numpy or tensors) -- clb_searcher_create_device."""
        import torch
        c, co, r, iv = index["centroids"], index["codes"], index["residuals"], index["ivf"]
        for t in (c, co, r, iv):
            if not (t.is_cuda and t.is_contiguous()):
                raise ColBERTError("device index arrays must be contiguous CUDA tensors")
        host = lambda a, dt: np.ascontiguousarray(a.cpu().numpy() if hasattr(a, "data_ptr") else a, dtype=dt)
        w, dl, il = host(index["bucket_weights"], np.float32), host(index["doclens"], np.int64), host(index["ivf_lengths"], np.int64)
        if c.dtype != torch.float32 or c.shape[1] != self.dim:
            raise ColBERTError("centroids must be a (K, dim) float32 tensor")
        if co.element_size() != 4 or r.dtype != torch.uint8 or iv.dtype != torch.int64:
            raise ColBERTError("codes must be 32-bit, residuals uint8, ivf int64")
        if tuple(r.shape) != (co.numel(), self.dim // 8 * self.nbits):
            raise ColBERTError("residuals must be (n_emb, dim/8*nbits)")
        if il.size != c.shape[0]:
            raise ColBERTError("ivf_lengths must have one entry per centroid")
        self.device = c.device.index
        self.num_centroids = int(c.shape[0]); self.num_docs = dl.size; self.num_embeddings = co.numel()
        torch.cuda.synchronize(c.device)
        self._h = C.c_void_p()
        check(lib().clb_searcher_create_device(self.device, i64(self.dim), C.c_int(self.nbits), i64(c.shape[0]),
                                               C.c_void_p(c.data_ptr()), fptr(w), i64(dl.size), fptr(dl), i64(co.numel()),
                                               C.c_void_p(co.data_ptr()), C.c_void_p(r.data_ptr()), C.c_void_p(iv.data_ptr()),
                                               fptr(il), i64(pid_offset), C.byref(self._h)))

    # -- lifetime ---------------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None):
            lib().clb_searcher_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def device_bytes(self) -> int:
        return int(lib().clb_searcher_device_bytes(self._h))

    def set_mode(self, mode: int):
        check(lib().clb_searcher_set_mode(self._h, C.c_int(mode)))

    def set_wide_select(self, on: int):
        """-1: the selection step picks one or sixteen work-groups per query by the candidate capacity; 0 / 1 force it."""
        check(lib().clb_searcher_set_wide_select(self._h, C.c_int(on)))

    def set_pass1_gather(self, form: int):
        """-1: pass 1 picks its gather form from the index's code statistics; 0 / 1 force the VGPR / LDS-DMA form."""
        check(lib().clb_searcher_set_pass1_gather(self._h, C.c_int(form)))

    @property
    def pass1_gather(self):
        """(form in use: 0 VGPR / 1 LDS-DMA, code adjacency statistic of the index)"""
        adj = C.c_double(0)
        return int(lib().clb_searcher_get_pass1_gather(self._h, C.byref(adj))), adj.value

    @property
    def mode(self) -> int:
        return int(lib().clb_searcher_get_mode(self._h))

    @property
    def bound_consts(self) -> np.ndarray:
        """The six constants of this handle's error bound (include/colbert_hip.h, clb_searcher_get_bound_consts)."""
        out = np.zeros(6, dtype=np.float32)
        check(lib().clb_searcher_get_bound_consts(self._h, fptr(out)))
        return out

    def raise_bound_consts(self, consts):
        """Element-wise maximum with `consts` (sharded search: one bound on every shard, distributed.sync_bound_consts)."""
        c = np.ascontiguousarray(consts, dtype=np.float32)
        assert c.shape == (6,)
        check(lib().clb_searcher_set_bound_consts(self._h, fptr(c)))

    # -- search -----------------------------------------------------------------------------------
    def search_embeddings(self, Q, k: int, nprobe: Optional[int] = None):
        """search() after encode_queries (searching.jl:102-127).  Q: (dim, T) Float32.
        Returns (pids Int64[k] 1-based, scores Float32[k])."""
        q = colmajor(Q, np.float32)
        if q.ndim != 2 or q.shape[0] != self.dim:
            raise ColBERTError(f"Q must be (dim={self.dim}, T)")
        pids = np.zeros(k, dtype=np.int64); scores = np.zeros(k, dtype=np.float32)
        ncand = i64(0)
        check(lib().clb_search(self._h, fptr(q), i64(q.shape[1]), i64(nprobe or self.config.nprobe), i64(k), fptr(pids),
                               fptr(scores), C.byref(ncand)))
        self.last_num_candidates = ncand.value
        return pids, scores

    def search_batch(self, Q, k: int, nprobe: Optional[int] = None, pad_short: bool = False):
        """B queries: Q (dim, T, B) -> (pids (k, B), scores (k, B), n_candidates[B])."""
        q = colmajor(Q, np.float32)
        if q.ndim != 3 or q.shape[0] != self.dim:
            raise ColBERTError(f"Q must be (dim={self.dim}, T, B)")
        B = q.shape[2]
        pids = np.zeros((k, B), dtype=np.int64, order="F"); scores = np.zeros((k, B), dtype=np.float32, order="F")
        ncand = np.zeros(B, dtype=np.int64)
        check(lib().clb_search_batch(self._h, fptr(q), i64(q.shape[1]), i64(B), i64(nprobe or self.config.nprobe), i64(k),
                                     C.c_int(1 if pad_short else 0), fptr(pids), fptr(scores), fptr(ncand)))
        return pids, scores, ncand

    def retrieve(self, Q, nprobe: Optional[int] = None):
        """retrieve (src/search/ranking.jl:23-44): ascending candidate pids."""
        q = colmajor(Q, np.float32)
        out = np.zeros(max(self.num_docs, 1), dtype=np.int64)
        n = i64(0)
        check(lib().clb_retrieve(self._h, fptr(q), i64(q.shape[1]), i64(nprobe or self.config.nprobe), fptr(out), C.byref(n)))
        return out[: n.value].copy()

    def debug_scores(self, Q, k: int, nprobe: Optional[int] = None) -> dict:
        """Two-pass test hook: candidates with approximate and exact scores, tau, eps, #re-scored."""
        q = colmajor(Q, np.float32)
        cap = max(self.num_docs, 1)
        pids = np.zeros(cap, dtype=np.int64); ap = np.zeros(cap, dtype=np.float32); ex = np.zeros(cap, dtype=np.float32)
        n = i64(0); nr = i64(0); tau = C.c_float(0); eps = C.c_float(0)
        check(lib().clb_debug_scores(self._h, fptr(q), i64(q.shape[1]), i64(nprobe or self.config.nprobe), i64(k), i64(cap),
                                     fptr(pids), fptr(ap), fptr(ex), C.byref(n), C.byref(tau), C.byref(eps), C.byref(nr)))
        m = n.value
        return {"pids": pids[:m].copy(), "approx": ap[:m].copy(), "exact": ex[:m].copy(), "tau": tau.value,
                "eps": eps.value, "n_rescore": nr.value}

    def search(self, query: str, k: int):
        """search(searcher, query::String, k) (searching.jl:93-128)."""
        if self.encoder is None:
            raise ColBERTError("no query encoder attached: pass encoder=... or use search_embeddings(Q, k)")
        Q = self.encoder.encode_queries([query])
        assert Q.shape[2] == 1 and Q.shape[1] == self.config.query_maxlen, Q.shape
        return self.search_embeddings(Q[:, :, 0], k)

    # -- profiling (bench.py) -----------------------------------------------------------------------
    def profile_enable(self, on: bool = True, counters: bool = False):
        """Per-kernel HIP-event timing; `counters` additionally fills `last_batch_stats` (one extra kernel per
        batch -- not for timed regions)."""
        check(lib().clb_profile_enable(self._h, C.c_int((2 if counters else 1) if on else 0)))

    def profile_read(self) -> dict:
        cap = 16
        names = (C.c_char_p * cap)(); ms = (C.c_double * cap)(); cnt = (C.c_int64 * cap)()
        n = lib().clb_profile_read(self._h, names, ms, cnt, cap)
        if n < 0:
            check(10)
        return {names[i].decode(): {"ms": ms[i], "launches": cnt[i]} for i in range(n)}

    def last_batch_stats(self) -> dict:
        a, b, c, d = i64(0), i64(0), i64(0), i64(0)
        check(lib().clb_last_batch_stats(self._h, C.byref(a), C.byref(b), C.byref(c), C.byref(d)))
        return {"cand_docs": a.value, "cand_embs": b.value, "rescored_docs": c.value, "rescored_embs": d.value}


def search(searcher: Searcher, query, k: int):
    """The reference's exported `search` (src/ColBERT.jl:40).  `query` may be a string (needs an
    encoder) or a (dim, T) Float32 matrix of query embeddings."""
    if isinstance(query, str):
        return searcher.search(query, k)
    return searcher.search_embeddings(query, k)
