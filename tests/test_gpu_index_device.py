"""The index build with every large array resident in HBM (clb_kmeans_shard_create_device, clb_codec_*,
clb_build_ivf_device, clb_searcher_create_device; indexer.index_device) against the CPU oracle and against the
host-buffer entry points.  Small shapes here; the same path at BASELINE's sizes is tests/test_gpu_sizes.py
(1/8 of config 5) and bench.py's `built_index_1M`.  Everything goes through the C ABI (ctypes)."""
import numpy as np
import pytest

import colbert_jl_amd as clb
from colbert_jl_amd import codec, indexer, synthetic

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def _torch():
    import torch
    return torch, torch.device("cuda", 0)


def test_kmeans_device_matches_oracle_and_host_entry_point(oracle):
    """Points borrowed in place from a CUDA tensor: same centroids (0 ulp) and assignments as clb_kmeans and the oracle."""
    torch, dev = _torch()
    rng = np.random.default_rng(5)
    n, K = 6000, 96
    x = rng.standard_normal((128, n)).astype(np.float32)
    x /= np.linalg.norm(x, axis=0, keepdims=True)
    x = np.asfortranarray(x)
    init = np.asfortranarray(x[:, rng.permutation(n)[:K]])
    rc, ra, rit = oracle.kmeans(x, init, max_iters=5)
    hc, ha, hit = codec.kmeans(x, init, max_iters=5)
    dx = torch.from_numpy(np.ascontiguousarray(x.T)).to(dev)
    dc, dit, sh = codec.kmeans_device(dx, torch.from_numpy(np.ascontiguousarray(init.T)).to(dev), max_iters=5)
    da = sh.get_assignments()
    sh.close()
    got = np.asfortranarray(dc.cpu().numpy().T)
    assert dit == rit == hit
    assert np.array_equal(bits(got), bits(rc)) and np.array_equal(bits(hc), bits(rc))
    assert np.array_equal(da, ra) and np.array_equal(ha, ra)


@pytest.mark.parametrize("nbits", [1, 2, 4])
def test_codec_compress_device_bit_exact(oracle, nbits):
    """Resident codec, two chunks into slices of one preallocated output: bytes identical to the oracle's compress."""
    torch, dev = _torch()
    rng = np.random.default_rng(11 + nbits)
    n, K = 5000, 300
    cent = np.asfortranarray((rng.standard_normal((128, K)) * 0.1).astype(np.float32))
    x = rng.standard_normal((128, n)).astype(np.float32)
    x /= np.linalg.norm(x, axis=0, keepdims=True)
    x = np.asfortranarray(x)
    cut = np.sort(rng.normal(0, 0.05, (1 << nbits) - 1).astype(np.float32))
    rc, rr = oracle.compress(cent, cut, 128, nbits, x)
    dx = torch.from_numpy(np.ascontiguousarray(x.T)).to(dev)
    codes = torch.empty(n, dtype=torch.int32, device=dev)
    res = torch.empty((n, 16 * nbits), dtype=torch.uint8, device=dev)
    cdc = codec.Codec(torch.from_numpy(np.ascontiguousarray(cent.T)).to(dev), cut, 128, nbits)
    cdc.compress_device(dx[:3001], codes[:3001], res[:3001])
    cdc.compress_device(dx[3001:], codes[3001:], res[3001:])
    torch.cuda.synchronize()
    cdc.close()
    assert np.array_equal(codes.cpu().numpy().view(np.uint32), rc)
    assert np.array_equal(res.cpu().numpy().T, rr)
    # host centroids into the same handle type
    cdc = codec.Codec(cent, cut, 128, nbits, device=0)
    c2, r2 = cdc.compress_device(dx)
    torch.cuda.synchronize()
    cdc.close()
    assert np.array_equal(c2.cpu().numpy().view(np.uint32), rc) and np.array_equal(r2.cpu().numpy().T, rr)


def test_build_ivf_device_matches_oracle(oracle):
    torch, dev = _torch()
    rng = np.random.default_rng(3)
    K = 1000
    codes = rng.integers(1, K + 1, size=200_000, dtype=np.uint32)
    codes[codes == 17] = 18                                              # an empty list
    rivf, rlens = oracle.build_ivf(codes, K)
    ivf, lens = codec.build_ivf_device(torch.from_numpy(codes.view(np.int32)).to(dev), K)
    assert np.array_equal(ivf.cpu().numpy(), rivf) and np.array_equal(lens.cpu().numpy(), rlens)
    bad = codes.copy(); bad[5] = K + 1
    with pytest.raises(clb.BoundsError):
        codec.build_ivf_device(torch.from_numpy(bad.view(np.int32)).to(dev), K)


def test_index_device_end_to_end(oracle):
    """index_device on 3 000 passages of the device mixture source: every stage re-derived by the oracle from the SAME
    inputs (the rng draws of index_device replayed on the host), then Searcher(index=device arrays) == Searcher(index=host
    arrays) == oracle.search."""
    torch, dev = _torch()
    n_docs = 3000
    src = synthetic.DeviceMixtureSource(seed=21, n_docs=n_docs, device=dev, block=700)
    # chunk() is reproducible and independent of the chunk boundaries
    a = src.chunk(0, n_docs)
    b = torch.cat([src.chunk(0, 650), src.chunk(650, 1500), src.chunk(1500, n_docs)])
    assert torch.equal(a, b)
    index, rec = indexer.index_device(src, nbits=2, kmeans_niters=3, chunksize=800, seed=9)
    embs = np.asfortranarray(a.cpu().numpy().T)
    doclens = src.doclens
    off = np.concatenate([[0], np.cumsum(doclens)])
    # replay of index_device's draws
    rng = np.random.default_rng(9)
    sampled = np.unique(rng.integers(0, n_docs, size=codec.num_sampled_pids(n_docs)))
    cols = np.concatenate([np.arange(off[p], off[p + 1]) for p in sampled])
    sample = embs[:, cols][:, rng.permutation(cols.size)]
    h = codec.heldout_size(sample.shape[1])
    sample, held = np.asfortranarray(sample[:, :-h]), np.asfortranarray(sample[:, -h:])
    K = rec["K"]
    assert K == codec.setup(n_docs, float(np.float32(doclens[sampled].sum() / sampled.size)), sample.shape[1], 800, 1)["num_partitions"]
    init = np.asfortranarray(sample[:, rng.permutation(sample.shape[1])[:K]])
    rcent, _, rit = oracle.kmeans(sample, init, max_iters=3)
    host = indexer.index_to_host(index)
    assert rec["kmeans_iters"] == rit
    assert np.array_equal(bits(host["centroids"]), bits(rcent))
    rcut, rw, ravg, _ = oracle.compute_avg_residuals(2, rcent, held)
    assert np.array_equal(bits(host["bucket_cutoffs"]), bits(rcut)) and np.array_equal(bits(host["bucket_weights"]), bits(rw))
    rc, rr = oracle.compress(rcent, rcut, 128, 2, embs)
    assert np.array_equal(host["codes"], rc) and np.array_equal(host["residuals"], rr)
    rivf, rlens = oracle.build_ivf(rc, K)
    assert np.array_equal(host["ivf"], rivf) and np.array_equal(host["ivf_lengths"], rlens)
    # search the built index: handle from the device arrays == handle from the host arrays == oracle
    Q = synthetic.make_queries(host, seed=4, n_queries=5)
    sd = clb.Searcher(index=index)
    sh = clb.Searcher(index=host, device=0)
    oidx = dict(host, emb2pid=oracle.build_emb2pid(host["doclens"]))
    for j in range(Q.shape[2]):
        rp, rs, _ = oracle.search(oidx, Q[:, :, j], 2, 100)
        for s in (sd, sh):
            p, sc = s.search_embeddings(Q[:, :, j], k=100)
            assert np.array_equal(p, rp) and np.array_equal(bits(sc), bits(rs))
    sd.close(); sh.close()


def test_gather_rows_and_plain_device_arrays():
    """clb_gather_rows_device (the sample / shuffle gathers of the device-resident build) against numpy, 16-byte and
    4-byte row pieces, into a slice of a larger buffer; an index outside the source is BoundsError.  clb_device_malloc /
    upload / download / memory: what a host without device arrays of its own (the Julia shim) drives the same route with."""
    import ctypes as C
    torch, dev = _torch()
    rng = np.random.default_rng(3)
    for cols in (128, 33):                       # 512-byte rows (uint4 pieces) and 132-byte rows (dword pieces)
        x = rng.standard_normal((1000, cols)).astype(np.float32)
        rows = rng.integers(0, 1000, size=2500)
        dx = torch.from_numpy(x).to(dev)
        got = codec.gather_rows_device(dx, rows)
        assert np.array_equal(got.cpu().numpy(), x[rows])
        big = torch.zeros((3000, cols), dtype=torch.float32, device=dev)
        codec.gather_rows_device(dx, rows[:700], out=big[100:800])
        b = big.cpu().numpy()
        assert np.array_equal(b[100:800], x[rows[:700]]) and not b[:100].any() and not b[800:].any()
    with pytest.raises(clb.BoundsError):
        codec.gather_rows_device(dx, np.array([0, 1000], dtype=np.int64))
    with pytest.raises(clb.BoundsError):
        codec.gather_rows_device(dx, np.array([-1], dtype=np.int64))
    # plain device arrays
    l = clb.lib()
    i64 = C.c_int64
    p = C.c_void_p()
    assert l.clb_device_malloc(0, i64(4096), C.byref(p)) == 0 and p.value
    src = rng.integers(0, 255, size=4096, dtype=np.uint8)
    back = np.zeros_like(src)
    assert l.clb_device_upload(0, p, src.ctypes.data_as(C.c_void_p), i64(4096)) == 0
    assert l.clb_device_download(0, back.ctypes.data_as(C.c_void_p), p, i64(4096)) == 0
    assert np.array_equal(src, back)
    assert l.clb_device_synchronize(0) == 0
    assert l.clb_device_free(0, p) == 0
    free, total = codec.device_memory(0)
    assert 0 < free <= total
