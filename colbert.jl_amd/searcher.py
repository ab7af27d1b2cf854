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

    def set_score_rows(self, form: int):
        """Batches of 16+ queries, two-pass mode: 0 = 64-byte fp16 score rows, 1 = 32-byte rows of 8-bit cells, -1 = default
        (0: the 8-bit format is faster in pass 1 only; set it alike on every shard of a group)."""
        check(lib().clb_searcher_set_score_rows(self._h, C.c_int(form)))

    @property
    def score_rows(self) -> int:
        """row format batches of 16+ queries take: 0 fp16 (64 B), 1 8-bit cells (32 B)"""
        return int(lib().clb_searcher_get_score_rows(self._h))

    def set_centroid_products(self, n: int):
        """Batches of 16+ queries, two-pass mode: 1 = score table from one fp16 product, 3 = the bf16 split, -1 = default (1 on a shard of a group, 3 on one GPU)."""
        check(lib().clb_searcher_set_centroid_products(self._h, C.c_int(n)))

    @property
    def centroid_products(self):
        """(products in use for batches of 16+ queries: 1 or 3, max ||c - fp16(c)|| over the centroids)"""
        dc = C.c_float(0)
        return int(lib().clb_searcher_get_centroid_products(self._h, C.byref(dc))), dc.value

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

    def text_search(self, k: int, nprobe: Optional[int] = None, graph: bool = True) -> "TextSearch":
        """A session for repeated `search(searcher, query::String, k)` calls with everything after the tokenizer on the
        device (TextSearch below)."""
        return TextSearch(self, k, nprobe, graph)

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


class TextSearch:
    """search(searcher, query::String, k) (src/searching.jl:93-128) for a serving loop: one text query per call --
    tokenize on the host, then encode_queries (1 x query_maxlen) -> search -> the k (pid, score) pairs, all on the device
    with no host round trip in between: the token ids go up in one 128-byte copy, the encoder writes the (1, T, dim) query
    tensor the search reads (clb_encode_queries_device -> clb_search_batch_device), the results come back in one copy.
    `graph=True`: the ~100 kernel launches of encoder + search are captured ONCE as a HIP graph over static buffers and
    replayed per query (the library only enqueues kernels and memsets on the stream it is handed).  Results are those of
    Searcher.search: BoundsError when the query has fewer than k candidates (searching.jl:127)."""

    def __init__(self, searcher: Searcher, k: int, nprobe: Optional[int] = None, graph: bool = True):
        import torch
        from .distributed import DeviceSearch
        if searcher.encoder is None or searcher.encoder.tokenizer is None:
            raise ColBERTError("text search needs an encoder with a tokenizer attached to the Searcher")
        self.s, self.enc, self.k = searcher, searcher.encoder, int(k)
        cfg = self.enc.config
        self.T = int(cfg.query_maxlen)
        self.dev = torch.device("cuda", searcher.device)
        tok = self.enc.tokenizer
        self.h_ids = torch.empty((1, self.T), dtype=torch.int32).pin_memory()
        self.h_mask = torch.empty((1, self.T), dtype=torch.uint8).pin_memory()
        self.d_ids = torch.zeros((1, self.T), dtype=torch.int32, device=self.dev)
        self.d_mask = torch.ones((1, self.T), dtype=torch.uint8, device=self.dev)
        self.d_skip = torch.tensor([tok.pad_id], dtype=torch.int64, device=self.dev)            # searching.jl:62
        self.d_q = torch.empty((1, self.T, searcher.dim), dtype=torch.float32, device=self.dev)
        self.run = DeviceSearch(searcher, self.T, 1, self.k, int(nprobe or searcher.config.nprobe))
        self.h_out = torch.empty(self.run.packed.numel(), dtype=torch.uint8).pin_memory()
        self.h_ncand = torch.empty(1, dtype=torch.int64).pin_memory()
        self.h_err = torch.zeros(1, dtype=torch.int32).pin_memory()
        # the encoder's sticky error flag (a 4-byte device word owned by the library) as a tensor view: copied back with every result
        flag = type("_Flag", (), {"__cuda_array_interface__": {"shape": (1,), "typestr": "<i4", "version": 2,
                                                                "data": (self.enc.error_flag_ptr(), False)}})()
        self.d_err = torch.as_tensor(flag, device=self.dev)
        self.stream = torch.cuda.Stream(self.dev)
        self.graph = None
        # a first pass sizes every workspace (allocations cannot be captured)
        self.d_ids.fill_(tok.lookup("[MASK]"))
        with torch.cuda.stream(self.stream):
            self._enqueue()
        self.stream.synchronize()
        self.enc.check_last_ids()
        if graph:
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph, stream=self.stream):
                self._enqueue()

    def _enqueue(self):
        self.enc.query_embeddings_device(self.d_ids, self.d_mask, self.d_skip, self.d_q)
        self.run(self.d_q)

    def __call__(self, query: str):
        """-> (pids Int64[k] 1-based, scores Float32[k])"""
        import torch
        from . import tokenization
        cfg = self.enc.config
        ids, mask = tokenization.tensorize_queries(cfg.query_token, cfg.attend_to_mask_tokens, self.enc.tokenizer, [query], self.T)
        self.h_ids.numpy()[0, :] = ids[:, 0]
        self.h_mask.numpy()[0, :] = mask[:, 0]
        with torch.cuda.stream(self.stream):
            self.d_ids.copy_(self.h_ids, non_blocking=True)
            self.d_mask.copy_(self.h_mask, non_blocking=True)
            if self.graph is not None:
                self.graph.replay()
            else:
                self._enqueue()
            self.h_out.copy_(self.run.packed, non_blocking=True)
            self.h_ncand.copy_(self.run.ncand, non_blocking=True)
            self.h_err.copy_(self.d_err, non_blocking=True)
        self.stream.synchronize()
        if int(self.h_err[0]) != 0:       # id outside the vocabulary / non-finite embeddings: raises (and clears the flag)
            self.enc.check_last_ids()
        n = int(self.h_ncand[0])
        self.s.last_num_candidates = n
        if n < self.k:                                                            # searching.jl:127
            raise BoundsError(f"attempt to access {n}-element Vector at index [1:{self.k}] (the query has {n} candidate passages)")
        k = self.k
        out = self.h_out.numpy()
        return out[:k * 8].view(np.int64).copy(), out[k * 8:k * 12].view(np.float32).copy()

    def search_many(self, queries, group: int = 128, batch: int = 32):
        """Many text queries at the throughput shape (round 6): encode_queries (src/modelling/checkpoint.jl:271-301) on `group`
        queries per call -- at 128 x 32 rows the encoder's Linear layers fill the chip, at 32 x 32 they do not -- and search
        (src/searching.jl:102-127) on `batch` queries per call, the size the search kernels are tuned for; the encode of group
        g + 1 runs on its own stream beside the later search batches of group g (it starts once the first of them has
        finished).  Returns [(pids, scores)] in query order, each identical to what __call__ returns for that query; raises
        BoundsError for the first query with fewer than k candidates."""
        import torch
        from . import tokenization
        from .distributed import DeviceSearch
        if group % batch != 0:
            raise ValueError("group must be a multiple of batch")
        n = len(queries)
        if n == 0:
            return []
        cfg = self.enc.config
        ids, mask = tokenization.tensorize_queries(cfg.query_token, cfg.attend_to_mask_tokens, self.enc.tokenizer, list(queries), self.T)
        n_groups = -(-n // group)
        pad = n_groups * group - n
        ids_t = np.ascontiguousarray(np.concatenate([ids.T] + [ids.T[-1:]] * pad, axis=0), dtype=np.int32)        # (n + pad, T)
        mask_t = np.ascontiguousarray(np.concatenate([mask.T] + [mask.T[-1:]] * pad, axis=0), dtype=np.uint8)
        st = getattr(self, "_many", None)
        if st is None or st["group"] != group or st["batch"] != batch:
            nprobe = self.run.nprobe
            st = {"group": group, "batch": batch,
                  "runs": [DeviceSearch(self.s, self.T, batch, self.k, nprobe, slot=1 + i) for i in range(2)],
                  "enc_stream": torch.cuda.Stream(self.dev), "compute": [torch.cuda.Stream(self.dev) for _ in range(2)],
                  "d_ids": [torch.zeros((group, self.T), dtype=torch.int32, device=self.dev) for _ in range(2)],
                  "d_mask": [torch.ones((group, self.T), dtype=torch.uint8, device=self.dev) for _ in range(2)],
                  "d_q": [torch.empty((group, self.T, self.s.dim), dtype=torch.float32, device=self.dev) for _ in range(2)]}
            self._many = st
        per = group // batch
        nb = n_groups * per
        h_ids = torch.from_numpy(ids_t).pin_memory()
        h_mask = torch.from_numpy(mask_t).pin_memory()
        h_out = torch.empty((nb, st["runs"][0].packed.numel()), dtype=torch.uint8).pin_memory()
        h_nc = torch.empty((nb, batch), dtype=torch.int64).pin_memory()
        gate = None
        done = [[], []]                    # events of the search batches that read d_q[0] / d_q[1]
        torch.cuda.current_stream(self.dev).synchronize()
        for g in range(n_groups):
            b2 = g % 2
            es = st["enc_stream"]
            with torch.cuda.stream(es):
                for ev in done[b2]:        # the group that used these buffers two groups ago has been searched
                    es.wait_event(ev)
                done[b2] = []
                st["d_ids"][b2].copy_(h_ids[g * group:(g + 1) * group], non_blocking=True)
                st["d_mask"][b2].copy_(h_mask[g * group:(g + 1) * group], non_blocking=True)
                if gate is not None:
                    es.wait_event(gate)
                self.enc.query_embeddings_device(st["d_ids"][b2], st["d_mask"][b2], self.d_skip, st["d_q"][b2])
                encoded = torch.cuda.Event()
                encoded.record(es)
            for j in range(per):
                i = g * per + j
                cs, run = st["compute"][i % 2], st["runs"][i % 2]
                with torch.cuda.stream(cs):
                    cs.wait_event(encoded)
                    run(st["d_q"][b2][j * batch:(j + 1) * batch])
                    h_out[i].copy_(run.packed, non_blocking=True)
                    h_nc[i].copy_(run.ncand, non_blocking=True)
                    ev = torch.cuda.Event()
                    ev.record(cs)
                    done[b2].append(ev)
                    if j == 0:
                        gate = ev
        with torch.cuda.stream(st["enc_stream"]):
            self.h_err.copy_(self.d_err, non_blocking=True)
        for cs in st["compute"]:
            cs.synchronize()
        st["enc_stream"].synchronize()
        if int(self.h_err[0]) != 0:
            self.enc.check_last_ids()
        k, out = self.k, []
        raw, nc = h_out.numpy(), h_nc.numpy()
        for q in range(n):
            i, r = divmod(q, batch)
            if int(nc[i, r]) < k:                                                  # searching.jl:127
                self.s.last_num_candidates = int(nc[i, r])
                raise BoundsError(f"attempt to access {int(nc[i, r])}-element Vector at index [1:{k}] (query {q + 1} has {int(nc[i, r])} candidate passages)")
            pids = raw[i, :batch * k * 8].view(np.int64).reshape(batch, k)[r].copy()
            scores = raw[i, batch * k * 8:batch * k * 12].view(np.float32).reshape(batch, k)[r].copy()
            out.append((pids, scores))
        self.s.last_num_candidates = int(nc[(n - 1) // batch, (n - 1) % batch])
        return out

    def close(self):
        self.graph = None
        self._many = None


def search(searcher: Searcher, query, k: int):
    """The reference's exported `search` (src/ColBERT.jl:40).  `query` may be a string (needs an
    encoder) or a (dim, T) Float32 matrix of query embeddings."""
    if isinstance(query, str):
        return searcher.search(query, k)
    return searcher.search_embeddings(query, k)
