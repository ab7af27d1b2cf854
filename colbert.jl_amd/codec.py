"""Host-side mirrors of the reference's codec / index-build functions over the C ABI.
Same names and argument meaning as src/indexing/codecs/residual.jl, src/utils.jl and
src/indexing/collection_indexer.jl; arrays keep the Julia shapes (column-major)."""
from __future__ import annotations

import ctypes as C
import math

import numpy as np

from ._lib import check, colmajor, fptr, i64, lib


def _normalize_array(X, dims: int = 1, device: int = 0):
    """_normalize_array!(X, dims=1)  (src/utils.jl:320-325); returns the normalised copy."""
    if dims != 1:
        raise NotImplementedError("only dims=1 runs on the device")
    X = colmajor(X, np.float32).copy(order="F")
    dim = X.shape[0]
    n = X.size // max(dim, 1)
    check(lib().clb_normalize_columns(device, fptr(X), i64(dim), i64(n)))
    return X


def compress_into_codes(centroids, embs, device: int = 0, n_codes=None):
    """compress_into_codes!  (residual.jl:67-81)"""
    c = colmajor(centroids, np.float32); x = colmajor(embs, np.float32)
    n = x.shape[1]
    codes = np.zeros(n if n_codes is None else n_codes, dtype=np.uint32)
    check(lib().clb_compress_into_codes(device, fptr(codes), i64(codes.size), fptr(c), i64(c.shape[0]),
                                        i64(c.shape[1]), fptr(x), i64(n)))
    return codes


def compress(centroids, bucket_cutoffs, dim: int, nbits: int, embs, device: int = 0):
    """compress  (residual.jl:586-604) -> (codes UInt32[n], residuals UInt8 (dim/8*nbits, n))"""
    c = colmajor(centroids, np.float32); cu = np.ascontiguousarray(bucket_cutoffs, dtype=np.float32)
    x = colmajor(embs, np.float32)
    n = x.shape[1]
    codes = np.zeros(n, dtype=np.uint32)
    res = np.zeros((max(dim // 8, 0) * nbits, n), dtype=np.uint8, order="F")
    check(lib().clb_compress(device, fptr(c), i64(c.shape[1]), fptr(cu), i64(cu.size), i64(dim), C.c_int(nbits),
                             fptr(x), i64(n), fptr(codes), fptr(res)))
    return codes, res


def decompress(dim: int, nbits: int, centroids, bucket_weights, codes, residuals, device: int = 0):
    """decompress  (residual.jl:759-784) -> Float32 (dim, n)"""
    c = colmajor(centroids, np.float32); w = np.ascontiguousarray(bucket_weights, dtype=np.float32)
    co = np.ascontiguousarray(codes, dtype=np.uint32); r = colmajor(residuals, np.uint8)
    out = np.zeros((dim, co.size), dtype=np.float32, order="F")
    check(lib().clb_decompress(device, i64(dim), C.c_int(nbits), fptr(c), i64(c.shape[1]), fptr(w), i64(w.size),
                               fptr(co), i64(co.size), fptr(r), i64(r.shape[0]), i64(r.shape[1]), fptr(out)))
    return out


def maxsim(Q, D, pids, doclens, device: int = 0):
    """maxsim  (src/search/ranking.jl:69-86)"""
    q = colmajor(Q, np.float32); d = colmajor(D, np.float32)
    p = np.ascontiguousarray(pids, dtype=np.int64); dl = np.ascontiguousarray(doclens, dtype=np.int64)
    scores = np.zeros(p.size, dtype=np.float32)
    check(lib().clb_maxsim(device, fptr(q), i64(q.shape[0]), i64(q.shape[1]), fptr(d),
                           i64(d.shape[1] if d.ndim == 2 else 0), fptr(p), i64(p.size), fptr(dl), i64(dl.size),
                           fptr(scores)))
    return scores


def kmeans(data, init_centroids, max_iters: int = 10, tol: float = 1e-4, point_bsize: int = 1000,
           device: int = 0):
    """kmeans_gpu_onehot!  (src/utils.jl:253-318) with the initial centroids injected.
    Returns (centroids, assignments Int32 1-based, iterations executed)."""
    x = colmajor(data, np.float32); c = colmajor(init_centroids, np.float32).copy(order="F")
    dim, n = x.shape
    assign = np.zeros(n, dtype=np.int32)
    iters = i64(0)
    check(lib().clb_kmeans(device, fptr(x), i64(dim), i64(n), fptr(c), i64(c.shape[1]), i64(max_iters),
                           C.c_float(tol), i64(point_bsize), fptr(assign), C.byref(iters)))
    return c, assign, iters.value


def _is_tensor(a) -> bool:
    return hasattr(a, "data_ptr") and hasattr(a, "is_cuda")


def _dev_rows(t, dtype=None):
    """A torch CUDA tensor as the device array the ABI takes: (n, dim) row-major IS the reference's (dim, n) column-major
    matrix.  Returns the (contiguous) tensor; the caller keeps it alive across the call."""
    import torch
    if not (t.is_cuda and t.is_contiguous()):
        raise ValueError("device arrays must be contiguous CUDA tensors")
    if dtype is not None and t.dtype != dtype:
        raise ValueError(f"expected {dtype}, got {t.dtype}")
    return t


def _stream_of(t):
    import torch
    return C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


class KMeansShard:
    """One rank's points of a sharded k-means (clb_kmeans_shard_*): uploaded once, one `pass_` per iteration.
    `data`: a (dim, n) numpy array (uploaded) or an (n, dim) float32 CUDA tensor (borrowed in place:
    clb_kmeans_shard_create_device -- the tensor is kept alive by this object)."""

    def __init__(self, data, K: int, point_bsize: int = 1000, device: int = 0):
        self._h = C.c_void_p()
        if _is_tensor(data):
            import torch
            x = _dev_rows(data, torch.float32)
            self._points = x                                    # borrowed by the handle
            self.n, self.dim = x.shape
            self.K, self.device = int(K), x.device.index
            check(lib().clb_kmeans_shard_create_device(self.device, C.c_void_p(x.data_ptr()), i64(self.dim), i64(self.n),
                                                       i64(K), i64(point_bsize), C.byref(self._h)))
            return
        x = colmajor(data, np.float32)
        self.dim, self.n = x.shape
        self.K, self.device = int(K), device
        check(lib().clb_kmeans_shard_create(device, fptr(x), i64(self.dim), i64(self.n), i64(K), i64(point_bsize),
                                            C.byref(self._h)))

    def pass_(self, centroids, want_assignments: bool = False):
        """-> (sums (dim,K) fp32, counts int64[K][, assignments int32 1-based])"""
        c = colmajor(centroids, np.float32)
        sums = np.zeros((self.dim, self.K), dtype=np.float32, order="F"); counts = np.zeros(self.K, dtype=np.int64)
        assign = np.zeros(self.n, dtype=np.int32) if want_assignments else None
        check(lib().clb_kmeans_shard_pass(self._h, fptr(c), fptr(sums), fptr(counts),
                                          fptr(assign) if want_assignments else None))
        return (sums, counts, assign) if want_assignments else (sums, counts)

    # -- the exchange on the device (torch CUDA tensors; enqueued on torch's current stream) ---------------------------
    @property
    def block_bytes(self) -> int:
        return int(lib().clb_kmeans_shard_block_bytes(self._h))

    def set_centroids(self, centroids):
        """(dim, K) numpy array or (K, dim) float32 CUDA tensor"""
        if _is_tensor(centroids):
            import torch
            c = _dev_rows(centroids, torch.float32)
            assert tuple(c.shape) == (self.K, self.dim)
            torch.cuda.current_stream(c.device).synchronize()
            check(lib().clb_kmeans_shard_set_centroids(self._h, C.c_void_p(c.data_ptr())))
            return
        c = colmajor(centroids, np.float32)
        assert c.shape == (self.dim, self.K)
        check(lib().clb_kmeans_shard_set_centroids(self._h, fptr(c)))

    def get_centroids(self, out=None):
        """-> (dim, K) numpy array, or into `out`, a (K, dim) float32 CUDA tensor"""
        if out is not None:
            import torch
            c = _dev_rows(out, torch.float32)
            assert tuple(c.shape) == (self.K, self.dim)
            torch.cuda.current_stream(c.device).synchronize()
            check(lib().clb_kmeans_shard_get_centroids(self._h, C.c_void_p(c.data_ptr())))
            return out
        c = np.zeros((self.dim, self.K), dtype=np.float32, order="F")
        check(lib().clb_kmeans_shard_get_centroids(self._h, fptr(c)))
        return c

    def get_assignments(self):
        """Int32[n], 1-based: the assignments of the last pass"""
        a = np.zeros(self.n, dtype=np.int32)
        check(lib().clb_kmeans_shard_get_assignments(self._h, fptr(a)))
        return a

    def pass_device(self, d_block):
        """this rank's [sums | counts] block into the uint8 CUDA tensor `d_block` (block_bytes long)"""
        import torch
        assert d_block.is_cuda and d_block.is_contiguous() and d_block.numel() * d_block.element_size() >= self.block_bytes
        st = torch.cuda.current_stream(d_block.device).cuda_stream
        check(lib().clb_kmeans_shard_pass_device(self._h, C.c_void_p(d_block.data_ptr()), C.c_void_p(st)))

    def update_device(self, d_gathered, world: int, tol: float = 1e-4):
        """rank-ordered reduction of the `world` gathered blocks + centroid update -> (delta, converged)"""
        import torch
        assert d_gathered.is_cuda and d_gathered.is_contiguous()
        assert d_gathered.numel() * d_gathered.element_size() >= world * self.block_bytes
        st = torch.cuda.current_stream(d_gathered.device).cuda_stream
        delta = C.c_float(0); conv = C.c_int(0)
        check(lib().clb_kmeans_shard_update_device(self._h, C.c_void_p(d_gathered.data_ptr()), i64(world), C.c_float(tol),
                                                   C.byref(delta), C.byref(conv), C.c_void_p(st)))
        return delta.value, bool(conv.value)

    def close(self):
        if getattr(self, "_h", None):
            lib().clb_kmeans_shard_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def kmeans_device(points, init_centroids, max_iters: int = 10, tol: float = 1e-4, point_bsize: int = 1000):
    """kmeans_gpu_onehot!  (src/utils.jl:253-318) on points that are already in HBM: `points` (n, dim) and
    `init_centroids` (K, dim) float32 CUDA tensors.  One shard, no exchange -- bit-identical to `kmeans` (the shard
    handle documents that).  Returns (centroids (K, dim) CUDA tensor, iterations executed, shard handle -- open, for
    `get_assignments`; close it)."""
    import torch
    sh = KMeansShard(points, init_centroids.shape[0], point_bsize)
    sh.set_centroids(init_centroids)
    blk = torch.empty(sh.block_bytes, dtype=torch.uint8, device=points.device)
    it = 0
    with torch.cuda.device(points.device):
        for it in range(1, max_iters + 1):
            sh.pass_device(blk)
            _delta, conv = sh.update_device(blk, 1, tol)
            if conv:
                break
    out = torch.empty_like(init_centroids)
    sh.get_centroids(out)
    return out, (it if max_iters > 0 else 0), sh


class Codec:
    """The chunk loop's resident codec (clb_codec_*): centroids and cutoffs uploaded once, `compress_device` per chunk
    of device-resident embeddings (compress, residual.jl:586-604)."""

    def __init__(self, centroids, bucket_cutoffs, dim: int, nbits: int, device: int = 0):
        self._h = C.c_void_p()
        self.dim, self.nbits = int(dim), int(nbits)
        cu = np.ascontiguousarray(bucket_cutoffs, dtype=np.float32)
        if _is_tensor(centroids):
            import torch
            c = _dev_rows(centroids, torch.float32)
            self.K, self.device = int(c.shape[0]), c.device.index
            torch.cuda.current_stream(c.device).synchronize()
            cptr = C.c_void_p(c.data_ptr())
        else:
            c = colmajor(centroids, np.float32)
            self.K, self.device = int(c.shape[1]), device
            cptr = fptr(c)
        check(lib().clb_codec_create(self.device, i64(dim), C.c_int(nbits), i64(self.K), cptr, fptr(cu), i64(cu.size),
                                     C.byref(self._h)))

    def compress_device(self, embs, out_codes=None, out_residuals=None):
        """embs (n, dim) float32 CUDA tensor -> (codes uint32-as-int32 [n] 1-based, residuals uint8 (n, dim/8*nbits)),
        enqueued on torch's current stream."""
        import torch
        x = _dev_rows(embs, torch.float32)
        n = x.shape[0]
        rows = self.dim // 8 * self.nbits
        codes = out_codes if out_codes is not None else torch.empty(n, dtype=torch.int32, device=x.device)
        res = out_residuals if out_residuals is not None else torch.empty((n, rows), dtype=torch.uint8, device=x.device)
        assert codes.is_contiguous() and res.is_contiguous() and codes.numel() == n and res.numel() == n * rows
        check(lib().clb_codec_compress_device(self._h, C.c_void_p(x.data_ptr()), i64(n), C.c_void_p(codes.data_ptr()),
                                              C.c_void_p(res.data_ptr()), _stream_of(x)))
        return codes, res

    def close(self):
        if getattr(self, "_h", None):
            lib().clb_codec_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def gather_rows_device(src, rows, out=None):
    """out[i, :] = src[rows[i], :] on the device (clb_gather_rows_device): the sampled / shuffled columns of the reference's
    (dim, n) matrices (`sample[:, randperm(...)]`, the sampled passages' embeddings -- collection_indexer.jl:56-91) cut out
    without torch indexing.  src: contiguous CUDA tensor (n_src, ...); rows: int64 CUDA tensor or numpy array, 0-based;
    out: optional contiguous destination (a slice of a larger buffer will do).  BoundsError for an index outside src."""
    import torch
    s_ = _dev_rows(src)
    if not _is_tensor(rows):
        rows = torch.from_numpy(np.ascontiguousarray(rows, dtype=np.int64)).to(s_.device)
    r_ = _dev_rows(rows, torch.int64)
    n = int(r_.numel())
    row_bytes = int(s_[0].numel() * s_.element_size()) if s_.shape[0] else int(np.prod(s_.shape[1:]) * s_.element_size())
    if out is None:
        out = torch.empty((n,) + tuple(s_.shape[1:]), dtype=s_.dtype, device=s_.device)
    o_ = _dev_rows(out, s_.dtype)
    assert o_.numel() * o_.element_size() == n * row_bytes, "gather_rows_device: destination size"
    check(lib().clb_gather_rows_device(s_.device.index, C.c_void_p(s_.data_ptr()), i64(s_.shape[0]), i64(row_bytes),
                                       C.c_void_p(r_.data_ptr()), i64(n), C.c_void_p(o_.data_ptr()), _stream_of(s_)))
    return out


def device_memory(device: int = 0):
    """(free, total) bytes of HBM on `device` (clb_device_memory)."""
    f, t = i64(0), i64(0)
    check(lib().clb_device_memory(device, C.byref(f), C.byref(t)))
    return int(f.value), int(t.value)


def build_ivf_device(codes, num_partitions: int):
    """_build_ivf  (collection_indexer.jl:349-353) over a device array: codes int32/uint32 CUDA tensor [n] (1-based) ->
    (ivf int64 CUDA tensor [n] 1-based, ivf_lengths int64 CUDA tensor [K])."""
    import torch
    co = _dev_rows(codes)
    assert co.element_size() == 4
    n = co.numel()
    ivf = torch.empty(n, dtype=torch.int64, device=co.device)
    lens = torch.empty(num_partitions, dtype=torch.int64, device=co.device)
    check(lib().clb_build_ivf_device(co.device.index, C.c_void_p(co.data_ptr()), i64(n), i64(num_partitions),
                                     C.c_void_p(ivf.data_ptr()), C.c_void_p(lens.data_ptr()), _stream_of(co)))
    return ivf, lens


def kmeans_reduce_update(centroids, gathered_sums, gathered_counts, tol: float = 1e-4, device: int = 0):
    """Rank-ordered reduction of the shards' partial sums + centroid update (clb_kmeans_reduce_update).
    gathered_sums: (world, dim*K) (each block a (dim,K) column-major matrix), gathered_counts: (world, K).
    Returns (centroids, delta, converged)."""
    c = colmajor(centroids, np.float32).copy(order="F")
    dim, K = c.shape
    gs = np.ascontiguousarray(np.asarray(gathered_sums, dtype=np.float32).reshape(-1, dim * K))
    gc = np.ascontiguousarray(np.asarray(gathered_counts, dtype=np.int64).reshape(gs.shape[0], K))
    delta = C.c_float(0); conv = C.c_int(0)
    check(lib().clb_kmeans_reduce_update(device, fptr(c), fptr(gs), fptr(gc), i64(gs.shape[0]), i64(dim), i64(K),
                                         C.c_float(tol), C.byref(delta), C.byref(conv)))
    return c, delta.value, bool(conv.value)


def compute_avg_residuals(nbits: int, centroids, heldout, device: int = 0, n_codes=None):
    """_compute_avg_residuals!  (collection_indexer.jl:177-195) -> (cutoffs, weights, avg_residual, codes)"""
    c = colmajor(centroids, np.float32); h = colmajor(heldout, np.float32)
    n = h.shape[1]
    codes = np.zeros(n if n_codes is None else n_codes, dtype=np.uint32)
    cut = np.zeros((1 << nbits) - 1, dtype=np.float32); w = np.zeros(1 << nbits, dtype=np.float32)
    avg = C.c_float(0)
    check(lib().clb_compute_avg_residuals(device, C.c_int(nbits), fptr(c), i64(c.shape[0]), i64(c.shape[1]),
                                          fptr(h), i64(n), fptr(codes), i64(codes.size), fptr(cut), fptr(w),
                                          C.byref(avg)))
    return cut, w, np.float32(avg.value), codes


def build_ivf(codes, num_partitions: int, device: int = 0):
    """_build_ivf  (collection_indexer.jl:349-353) -> (ivf Int64[n] 1-based, ivf_lengths Int64[K])"""
    co = np.ascontiguousarray(codes, dtype=np.uint32)
    ivf = np.zeros(co.size, dtype=np.int64); lens = np.zeros(num_partitions, dtype=np.int64)
    check(lib().clb_build_ivf(device, fptr(co), i64(co.size), i64(num_partitions), fptr(ivf), fptr(lens)))
    return ivf, lens


def doc_epilogue(D, integer_ids, skiplist, device: int = 0):
    """_doc_embeddings_and_doclens after the encoder forward  (checkpoint.jl:30-51)"""
    Dm = colmajor(D, np.float32)
    dim, L, N = Dm.shape
    ids = colmajor(integer_ids, np.int32); sk = np.ascontiguousarray(skiplist, dtype=np.int64)
    out = np.zeros((dim, max(L * N, 1)), dtype=np.float32, order="F")
    doclens = np.zeros(N, dtype=np.int64)
    n_out = i64(0)
    check(lib().clb_doc_epilogue(device, fptr(Dm), i64(dim), i64(L), i64(N), fptr(ids), fptr(sk), i64(sk.size),
                                 fptr(out), fptr(doclens), C.byref(n_out)))
    return np.asfortranarray(out[:, : n_out.value]), doclens


def query_epilogue(Q, integer_ids, skiplist, device: int = 0):
    """_query_embeddings after the encoder forward  (checkpoint.jl:61-69)"""
    Qm = colmajor(Q, np.float32).copy(order="F")
    dim, L, N = Qm.shape
    ids = colmajor(integer_ids, np.int32); sk = np.ascontiguousarray(skiplist, dtype=np.int64)
    check(lib().clb_query_epilogue(device, fptr(Qm), i64(dim), i64(L), i64(N), fptr(ids), fptr(sk), i64(sk.size)))
    return Qm


# ---- host-only planning helpers (collection_indexer.jl:17-24, 81-91, 115-139, 342-347) --------------
def num_sampled_pids(num_documents: int) -> int:
    v = 16 * math.sqrt(120 * num_documents)
    return int(min(1 + math.floor(v), num_documents))


def heldout_size(num_sample_embs: int, heldout_fraction: float = 0.05) -> int:
    prod = np.float32(heldout_fraction) * np.float32(num_sample_embs)
    return int(max(1, math.floor(min(np.float32(50000), prod))))


def setup(num_documents: int, avg_doclen_est: float, num_clustering_embs: int, chunksize, nranks: int) -> dict:
    if chunksize is None:
        chunksize = min(25000, 1 + num_documents // nranks)
    num_chunks = -(-num_documents // chunksize)
    est = np.float32(num_documents) * np.float32(avg_doclen_est)
    parts = math.floor(2 ** math.floor(math.log2(float(np.float32(16) * np.sqrt(est)))))
    return {"chunksize": int(chunksize), "num_chunks": int(num_chunks),
            "num_partitions": int(min(num_clustering_embs, parts)), "num_documents": int(num_documents),
            "num_embeddings_est": float(est), "avg_doclen_est": float(np.float32(avg_doclen_est))}


def collect_embedding_id_offset(chunk_emb_counts):
    cnt = list(chunk_emb_counts)
    if not cnt:
        return 0, np.zeros(1, dtype=np.int64)
    off = np.cumsum([1] + cnt[:-1]).astype(np.int64)
    return int(sum(cnt)), off
