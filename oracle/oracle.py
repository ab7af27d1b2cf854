"""ctypes front-end of the CPU oracle (oracle/colbert_oracle.c).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the
product package (colbert.jl_amd) never does.  Arrays keep the *Julia* shapes of the reference --
a matrix the reference calls (dim, n) is a numpy array of shape (dim, n) -- and are handed to C in
column-major (Fortran) order, which is how a Julia Array is laid out.  Ids are 1-based.

Errors: the C functions return the ORC_E* code of the Julia exception the reference would throw;
they are re-raised here as the Python classes below so tests read like the reference's tests.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libcolbert_oracle.so")


class DimensionMismatch(Exception):
    pass


class DomainError(Exception):
    pass


class BoundsError(Exception):
    pass


class ArgumentError(Exception):
    pass


_ERR = {1: DimensionMismatch, 2: DomainError, 3: BoundsError, 4: ArgumentError}


def build(force: bool = False) -> str:
    """Compile the oracle with gcc (oracle/Makefile).  Building the checker is not using it."""
    src = os.path.join(_HERE, "colbert_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.run(["make", "-C", _HERE, "-B", "libcolbert_oracle.so"], check=True,
                       stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.orc_dot.restype = C.c_float
        _lib.orc_sumsq.restype = C.c_float
        _lib.orc_num_sampled_pids.restype = C.c_int64
        _lib.orc_heldout_size.restype = C.c_int64
    return _lib


def _chk(rc: int) -> None:
    if rc != 0:
        raise _ERR.get(rc, RuntimeError)(f"oracle error code {rc}")


def _f(a, dtype):
    return np.asfortranarray(np.asarray(a, dtype=dtype))


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


i64 = C.c_int64


# ---- src/utils.jl ---------------------------------------------------------------------------------
def dot(a, b) -> np.float32:
    a = _f(a, np.float32).ravel(); b = _f(b, np.float32).ravel()
    return np.float32(lib().orc_dot(_p(a), _p(b), i64(a.size)))


def sumsq(x) -> np.float32:
    x = _f(x, np.float32).ravel()
    return np.float32(lib().orc_sumsq(_p(x), i64(x.size)))


def normalize_array(X, dims: int = 1):
    X = _f(X, np.float32).copy(order="F")
    if X.ndim == 1:
        X = X.reshape(-1, 1, order="F")
    if dims == 1:
        rows = X.shape[0]
        _chk(lib().orc_normalize_columns(_p(X), i64(rows), i64(X.size // max(rows, 1))))
    else:
        _chk(lib().orc_normalize_rows(_p(X), i64(X.shape[0]), i64(X.shape[1])))
    return X


def topk(data, k: int, dims: int = 1):
    data = _f(data, np.float32)
    rows, cols = data.shape
    if dims not in (1, 2):
        raise DomainError("dims must be 1 or 2!")
    out = np.zeros((rows, k) if dims == 2 else (k, cols), dtype=np.int64, order="F")
    _chk(lib().orc_topk(_p(data), i64(rows), i64(cols), i64(k), C.c_int(dims), _p(out)))
    return out


def compute_distances_kernel(batch_distances, batch_data, centroids):
    bd = _f(batch_distances, np.float32)
    x = _f(batch_data, np.float32); c = _f(centroids, np.float32)
    out = np.zeros(bd.shape, dtype=np.float32, order="F")
    _chk(lib().orc_kmeans_distances(_p(out), i64(bd.shape[0]), i64(bd.shape[1]), _p(x),
                                    i64(x.shape[0]), i64(x.shape[1]), _p(c), i64(c.shape[0]),
                                    i64(c.shape[1])))
    return out


def assign_clusters_kernel(n_assign: int, batch_distances):
    d = _f(batch_distances, np.float32)
    out = np.zeros(n_assign, dtype=np.int32)
    _chk(lib().orc_kmeans_assign(_p(out), i64(n_assign), _p(d), i64(d.shape[0]), i64(d.shape[1])))
    return out


def onehot_encode(batch_one_hot, batch_assignments, k: int):
    oh = _f(batch_one_hot, np.float32).copy(order="F")
    a = np.ascontiguousarray(batch_assignments, dtype=np.int32)
    _chk(lib().orc_onehot_encode(_p(oh), i64(oh.shape[0]), i64(oh.shape[1]), _p(a), i64(a.size), i64(k)))
    return oh


def update_centroids_kernel(new_centroids, batch_data, batch_one_hot):
    nc = _f(new_centroids, np.float32).copy(order="F")
    x = _f(batch_data, np.float32); oh = _f(batch_one_hot, np.float32)
    _chk(lib().orc_kmeans_update(_p(nc), i64(nc.shape[0]), i64(nc.shape[1]), _p(x), i64(x.shape[0]),
                                 i64(x.shape[1]), _p(oh), i64(oh.shape[0]), i64(oh.shape[1])))
    return nc


def kmeans(data, init_centroids, max_iters: int = 10, tol: float = 1e-4, point_bsize: int = 1000):
    """kmeans_gpu_onehot! with the initial centroids injected.  Returns (centroids, assignments, iters)."""
    x = _f(data, np.float32); c = _f(init_centroids, np.float32).copy(order="F")
    dim, n = x.shape
    if c.shape[0] != dim:
        raise DimensionMismatch("centroids and data must share the embedding dimension")
    assign = np.zeros(n, dtype=np.int32)
    iters = i64(0)
    _chk(lib().orc_kmeans(_p(x), i64(dim), i64(n), _p(c), i64(c.shape[1]), i64(max_iters),
                          C.c_float(tol), i64(point_bsize), _p(assign), C.byref(iters)))
    return c, assign, iters.value


def kmeans_shard_pass(data, centroids, point_bsize: int = 1000):
    """One iteration's pass over one shard's points -> (sums (dim,K) fp32, counts int64[K], assignments int32 1-based)."""
    x = _f(data, np.float32); c = _f(centroids, np.float32)
    dim, n = x.shape
    K = c.shape[1]
    sums = np.zeros((dim, K), dtype=np.float32, order="F"); counts = np.zeros(K, dtype=np.int64)
    assign = np.zeros(n, dtype=np.int32)
    _chk(lib().orc_kmeans_shard_pass(_p(x), i64(dim), i64(n), _p(c), i64(K), i64(point_bsize), _p(sums), _p(counts),
                                     _p(assign)))
    return sums, counts, assign


def kmeans_reduce_update(centroids, gathered_sums, gathered_counts, tol: float = 1e-4):
    """Rank-ordered reduction of the shards' partial sums + the centroid update.  gathered_sums: (world, K*dim)
    or (world, dim, K)-shaped, gathered_counts: (world, K).  Returns (centroids, delta, converged)."""
    c = _f(centroids, np.float32).copy(order="F")
    dim, K = c.shape
    gs = np.ascontiguousarray(np.stack([np.asfortranarray(np.asarray(g, dtype=np.float32).reshape(dim, K, order="F")).ravel(order="F")
                                        for g in gathered_sums]))
    gc = np.ascontiguousarray(np.asarray(gathered_counts, dtype=np.int64).reshape(len(gs), K))
    delta = C.c_float(0); conv = C.c_int(0)
    _chk(lib().orc_kmeans_reduce_update(_p(c), _p(gs), _p(gc), i64(len(gs)), i64(dim), i64(K), C.c_float(tol),
                                        C.byref(delta), C.byref(conv)))
    return c, delta.value, bool(conv.value)


# ---- src/indexing/codecs/residual.jl -----------------------------------------------------------------
def compress_into_codes(centroids, embs, n_codes=None):
    c = _f(centroids, np.float32); x = _f(embs, np.float32)
    n = x.shape[1]
    codes = np.zeros(n if n_codes is None else n_codes, dtype=np.uint32)
    _chk(lib().orc_compress_into_codes(_p(codes), i64(codes.size), _p(c), i64(c.shape[0]),
                                       i64(c.shape[1]), _p(x), i64(n)))
    return codes


def binarize_bits(data, nbits: int):
    d = _f(data, np.int64)
    dim, b = d.shape
    bits = np.zeros((nbits, dim, b), dtype=np.uint8, order="F")
    _chk(lib().orc_binarize_bits(_p(d), i64(dim), i64(b), C.c_int(nbits), _p(bits)))
    return bits.astype(bool)


def unbinarize(bits):
    b3 = _f(np.asarray(bits).astype(np.uint8), np.uint8)
    nbits, dim, b = b3.shape
    out = np.zeros((dim, b), dtype=np.int64, order="F")
    _chk(lib().orc_unbinarize(_p(b3), C.c_int(nbits), i64(dim), i64(b), _p(out)))
    return out


def bucket_indices(data, cutoffs):
    d = _f(data, np.float32); cu = np.ascontiguousarray(cutoffs, dtype=np.float32)
    out = np.zeros(d.shape, dtype=np.int64, order="F")
    _chk(lib().orc_bucket_indices(_p(d), i64(d.size), _p(cu), i64(cu.size), _p(out)))
    return out


def packbits(bits):
    b3 = _f(np.asarray(bits).astype(np.uint8), np.uint8)
    nbits, dim, b = b3.shape
    if dim % 8 != 0:
        raise DomainError("dim should be a multiple of 8!")
    out = np.zeros(((dim >> 3) * nbits, b), dtype=np.uint8, order="F")
    _chk(lib().orc_packbits(_p(b3), C.c_int(nbits), i64(dim), i64(b), _p(out)))
    return out


def unpackbits(packed, nbits: int):
    pk = _f(packed, np.uint8)
    rows, b = pk.shape
    if rows % nbits != 0:
        raise DomainError("first dimension must be a multiple of nbits")
    dim = (rows // nbits) << 3
    bits = np.zeros((nbits, dim, b), dtype=np.uint8, order="F")
    _chk(lib().orc_unpackbits(_p(pk), i64(rows), i64(b), C.c_int(nbits), _p(bits)))
    return bits.astype(bool)


def binarize(dim: int, nbits: int, cutoffs, residuals):
    cu = np.ascontiguousarray(cutoffs, dtype=np.float32); r = _f(residuals, np.float32)
    out = np.zeros((max(dim // 8, 0) * nbits, r.shape[1]), dtype=np.uint8, order="F")
    _chk(lib().orc_binarize(i64(dim), C.c_int(nbits), _p(cu), i64(cu.size), _p(r), i64(r.shape[1]), _p(out)))
    return out


def compress(centroids, cutoffs, dim: int, nbits: int, embs):
    c = _f(centroids, np.float32); cu = np.ascontiguousarray(cutoffs, dtype=np.float32)
    x = _f(embs, np.float32)
    n = x.shape[1]
    codes = np.zeros(n, dtype=np.uint32)
    res = np.zeros((dim // 8 * nbits, n), dtype=np.uint8, order="F")
    _chk(lib().orc_compress(_p(c), i64(c.shape[1]), _p(cu), i64(cu.size), i64(dim), C.c_int(nbits),
                            _p(x), i64(n), _p(codes), _p(res)))
    return codes, res


def decompress_residuals(dim: int, nbits: int, weights, packed):
    w = np.ascontiguousarray(weights, dtype=np.float32); pk = _f(packed, np.uint8)
    out = np.zeros((dim, pk.shape[1]), dtype=np.float32, order="F")
    _chk(lib().orc_decompress_residuals(i64(dim), C.c_int(nbits), _p(w), i64(w.size), _p(pk),
                                        i64(pk.shape[0]), i64(pk.shape[1]), _p(out)))
    return out


def decompress(dim: int, nbits: int, centroids, weights, codes, residuals):
    c = _f(centroids, np.float32); w = np.ascontiguousarray(weights, dtype=np.float32)
    co = np.ascontiguousarray(codes, dtype=np.uint32); r = _f(residuals, np.uint8)
    out = np.zeros((dim, co.size), dtype=np.float32, order="F")
    _chk(lib().orc_decompress(i64(dim), C.c_int(nbits), _p(c), i64(c.shape[1]), _p(w), i64(w.size),
                              _p(co), i64(co.size), _p(r), i64(r.shape[0]), i64(r.shape[1]), _p(out)))
    return out


# ---- src/indexing/collection_indexer.jl ------------------------------------------------------------
def num_sampled_pids(num_documents: int) -> int:
    return int(lib().orc_num_sampled_pids(i64(num_documents)))


def heldout_size(num_sample_embs: int, heldout_fraction: float = 0.05) -> int:
    return int(lib().orc_heldout_size(i64(num_sample_embs), C.c_float(heldout_fraction)))


def setup(num_documents: int, avg_doclen_est: float, num_clustering_embs: int, chunksize, nranks: int):
    cs, nc, npart = i64(0), i64(0), i64(0)
    est = C.c_double(0)
    _chk(lib().orc_setup(i64(num_documents), C.c_float(avg_doclen_est), i64(num_clustering_embs),
                         i64(-1 if chunksize is None else chunksize), i64(nranks), C.byref(cs),
                         C.byref(nc), C.byref(npart), C.byref(est)))
    return {"chunksize": cs.value, "num_chunks": nc.value, "num_partitions": npart.value,
            "num_documents": num_documents, "num_embeddings_est": est.value,
            "avg_doclen_est": float(np.float32(avg_doclen_est))}


def bucket_cutoffs_and_weights(nbits: int, heldout_avg_residual):
    v = np.ascontiguousarray(np.asarray(heldout_avg_residual, dtype=np.float32).ravel(order="F")).copy()
    cut = np.zeros((1 << nbits) - 1, dtype=np.float32); w = np.zeros(1 << nbits, dtype=np.float32)
    _chk(lib().orc_bucket_cutoffs_and_weights(C.c_int(nbits), _p(v), i64(v.size), _p(cut), _p(w)))
    return cut, w


def compute_avg_residuals(nbits: int, centroids, heldout, n_codes=None):
    c = _f(centroids, np.float32); h = _f(heldout, np.float32)
    n = h.shape[1]
    codes = np.zeros(n if n_codes is None else n_codes, dtype=np.uint32)
    cut = np.zeros((1 << nbits) - 1, dtype=np.float32); w = np.zeros(1 << nbits, dtype=np.float32)
    avg = C.c_float(0)
    _chk(lib().orc_compute_avg_residuals(C.c_int(nbits), _p(c), i64(c.shape[0]), i64(c.shape[1]),
                                         _p(h), i64(n), _p(codes), i64(codes.size), _p(cut), _p(w),
                                         C.byref(avg)))
    return cut, w, np.float32(avg.value), codes


def collect_embedding_id_offset(chunk_emb_counts):
    cnt = np.ascontiguousarray(chunk_emb_counts, dtype=np.int64)
    off = np.zeros(max(cnt.size, 1), dtype=np.int64)
    tot = i64(0)
    _chk(lib().orc_collect_embedding_id_offset(_p(cnt), i64(cnt.size), C.byref(tot), _p(off)))
    return tot.value, off


def build_ivf(codes, num_partitions: int):
    co = np.ascontiguousarray(codes, dtype=np.uint32)
    ivf = np.zeros(co.size, dtype=np.int64); lens = np.zeros(num_partitions, dtype=np.int64)
    _chk(lib().orc_build_ivf(_p(co), i64(co.size), i64(num_partitions), _p(ivf), _p(lens)))
    return ivf, lens


# ---- encoder epilogue ---------------------------------------------------------------------------
def doc_epilogue(D, integer_ids, skiplist):
    Dm = _f(D, np.float32).copy(order="F")
    dim, L, N = Dm.shape
    ids = _f(integer_ids, np.int32); sk = np.ascontiguousarray(skiplist, dtype=np.int64)
    out = np.zeros((dim, L * N), dtype=np.float32, order="F")
    doclens = np.zeros(N, dtype=np.int64)
    n_out = i64(0)
    _chk(lib().orc_doc_epilogue(_p(Dm), i64(dim), i64(L), i64(N), _p(ids), _p(sk), i64(sk.size),
                                _p(out), _p(doclens), C.byref(n_out)))
    return np.asfortranarray(out[:, : n_out.value]), doclens


def query_epilogue(Q, integer_ids, skiplist):
    Qm = _f(Q, np.float32).copy(order="F")
    dim, L, N = Qm.shape
    ids = _f(integer_ids, np.int32); sk = np.ascontiguousarray(skiplist, dtype=np.int64)
    _chk(lib().orc_query_epilogue(_p(Qm), i64(dim), i64(L), i64(N), _p(ids), _p(sk), i64(sk.size)))
    return Qm


# ---- src/search/ranking.jl, src/searching.jl -------------------------------------------------------
def build_emb2pid(doclens):
    dl = np.ascontiguousarray(doclens, dtype=np.int64)
    out = np.zeros(int(dl.sum()), dtype=np.int64)
    _chk(lib().orc_build_emb2pid(_p(dl), i64(dl.size), _p(out)))
    return out


def cids_to_eids(n_eids: int, centroid_ids, ivf, ivf_lengths):
    cids = np.ascontiguousarray(centroid_ids, dtype=np.int64)
    iv = np.ascontiguousarray(ivf, dtype=np.int64); il = np.ascontiguousarray(ivf_lengths, dtype=np.int64)
    eids = np.zeros(n_eids, dtype=np.int64)
    _chk(lib().orc_cids_to_eids(_p(eids), i64(n_eids), _p(cids), i64(cids.size), _p(iv), i64(iv.size),
                                _p(il), i64(il.size)))
    return eids


def retrieve(ivf, ivf_lengths, centroids, emb2pid, nprobe: int, Q):
    iv = np.ascontiguousarray(ivf, dtype=np.int64); il = np.ascontiguousarray(ivf_lengths, dtype=np.int64)
    c = _f(centroids, np.float32); e2p = np.ascontiguousarray(emb2pid, dtype=np.int64)
    q = _f(Q, np.float32)
    out = np.zeros(max(e2p.size, 1), dtype=np.int64)
    n_out = i64(0)
    _chk(lib().orc_retrieve(_p(iv), i64(iv.size), _p(il), i64(il.size), _p(c), i64(c.shape[0]), _p(e2p),
                            i64(e2p.size), i64(nprobe), _p(q), i64(q.shape[1]), _p(out), C.byref(n_out)))
    return out[: n_out.value].copy()


def collect_compressed_embs_for_pids(doclens, codes, residuals, pids):
    dl = np.ascontiguousarray(doclens, dtype=np.int64); co = np.ascontiguousarray(codes, dtype=np.uint32)
    r = _f(residuals, np.uint8); p = np.ascontiguousarray(pids, dtype=np.int64)
    n = int(dl[p - 1].sum()) if p.size else 0
    oc = np.zeros(n, dtype=np.uint32); orr = np.zeros((r.shape[0], n), dtype=np.uint8, order="F")
    _chk(lib().orc_collect_compressed(_p(dl), i64(dl.size), _p(co), _p(r), i64(r.shape[0]), _p(p),
                                      i64(p.size), _p(oc), _p(orr)))
    return oc, orr


def maxsim(Q, D, pids, doclens):
    q = _f(Q, np.float32); d = _f(D, np.float32)
    p = np.ascontiguousarray(pids, dtype=np.int64); dl = np.ascontiguousarray(doclens, dtype=np.int64)
    scores = np.zeros(p.size, dtype=np.float32)
    _chk(lib().orc_maxsim(_p(q), i64(q.shape[0]), i64(q.shape[1]), _p(d), i64(d.shape[1] if d.ndim == 2 else 0),
                          _p(p), i64(p.size), _p(dl), i64(dl.size), _p(scores)))
    return scores


def search(index: dict, Q, nprobe: int, k: int):
    """search() after the encoder (searching.jl:102-127).  `index` holds the Searcher's arrays:
    dim, nbits, centroids (dim,K), bucket_weights, doclens, codes, residuals (rows,n_emb), ivf,
    ivf_lengths, emb2pid (optional).  Returns (pids[k], scores[k], n_candidates)."""
    c = _f(index["centroids"], np.float32); w = np.ascontiguousarray(index["bucket_weights"], dtype=np.float32)
    dl = np.ascontiguousarray(index["doclens"], dtype=np.int64)
    co = np.ascontiguousarray(index["codes"], dtype=np.uint32); r = _f(index["residuals"], np.uint8)
    iv = np.ascontiguousarray(index["ivf"], dtype=np.int64)
    il = np.ascontiguousarray(index["ivf_lengths"], dtype=np.int64)
    e2p = index.get("emb2pid")
    e2p = build_emb2pid(dl) if e2p is None else np.ascontiguousarray(e2p, dtype=np.int64)
    q = _f(Q, np.float32)
    pids = np.zeros(max(k, 1), dtype=np.int64); scores = np.zeros(max(k, 1), dtype=np.float32)
    ncand = i64(0)
    _chk(lib().orc_search(i64(c.shape[0]), C.c_int(int(index["nbits"])), i64(c.shape[1]), _p(c), _p(w),
                          i64(dl.size), _p(dl), i64(co.size), _p(co), _p(r), _p(iv), _p(il), _p(e2p), _p(q),
                          i64(q.shape[1]), i64(nprobe), i64(k), _p(pids), _p(scores), C.byref(ncand)))
    return pids[:k], scores[:k], ncand.value


def num_threads() -> int:
    return int(lib().orc_num_threads())


def set_num_threads(n: int) -> None:
    lib().orc_set_num_threads(C.c_int(n))
