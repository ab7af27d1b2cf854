"""Parity tests proper: the HIP path, called through the C ABI (libcolbert_hip.so), against the CPU
oracle on the same seeded inputs, against the reference's golden vectors, and -- at BASELINE.json's
sizes -- through size-independent properties.  Bar: pids bit-exact, fp32 scores bit-exact against the
oracle's canonical arithmetic (tolerance 0; north_star allows 1e-4), bytes/indices bit-exact."""
import os

import numpy as np
import pytest

import colbert_jl_amd as clb
from colbert_jl_amd import codec, synthetic

pytestmark = pytest.mark.gpu

SCORE_TOL = 0.0   # fp32 scores are compared bit-for-bit; north_star's bound is 1e-4


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def assert_same_f32(a, b, what=""):
    a = np.asarray(a, dtype=np.float32); b = np.asarray(b, dtype=np.float32)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    if not np.array_equal(bits(a), bits(b)):
        diff = np.abs(a.astype(np.float64) - b.astype(np.float64))
        raise AssertionError(f"{what}: {np.count_nonzero(bits(a) != bits(b))} of {a.size} differ, max |d| = {diff.max()}")


def pad_dim(a, dim=128):
    a = np.asarray(a, dtype=np.float32)
    out = np.zeros((dim,) + a.shape[1:], dtype=np.float32, order="F")
    out[: a.shape[0]] = a
    return out


# ---------------------------------------------------------------------------------------------------
# golden vectors of the reference's tests, through the C ABI
# ---------------------------------------------------------------------------------------------------
def test_golden_maxsim(kats):
    k = kats["maxsim"]
    got = codec.maxsim(np.array(k["Q"], np.float32), np.array(k["D"], np.float32), k["pids"], k["doclens"])
    assert np.array_equal(got, np.array(k["expected_scores"], np.float32))


def test_golden_retrieve(kats):
    """test/search/ranking.jl:74-82, embedded in dim 128 by zero padding (dot products unchanged)."""
    k = kats["retrieve"]
    emb2pid = np.array(k["emb2pid"])
    n_docs = int(emb2pid.max())
    doclens = np.bincount(emb2pid, minlength=n_docs + 1)[1:]
    n_emb = emb2pid.size
    idx = {"dim": 128, "nbits": 2, "centroids": pad_dim(np.array(k["centroids"], np.float32)),
           "bucket_weights": synthetic.README_BUCKET_WEIGHTS, "doclens": doclens,
           "codes": np.ones(n_emb, np.uint32), "residuals": np.zeros((32, n_emb), np.uint8),
           "ivf": np.array(k["ivf"]), "ivf_lengths": np.array(k["ivf_lengths"])}
    s = clb.Searcher(index=idx)
    got = s.retrieve(pad_dim(np.array(k["Q"], np.float32)), nprobe=k["nprobe"])
    assert np.array_equal(got, k["expected_pids"])
    s.close()


def test_golden_build_ivf(kats):
    k = kats["_build_ivf"]
    ivf, lens = codec.build_ivf(k["codes"], k["num_partitions"])
    assert np.array_equal(ivf, k["expected_ivf"]) and np.array_equal(lens, k["expected_ivf_lengths"])


def test_golden_bucket_cutoffs(kats):
    """_bucket_cutoffs_and_weights KAT via _compute_avg_residuals! with a zero centroid (residual = data)."""
    k = kats["_bucket_cutoffs_and_weights"]
    held = np.array(k["heldout_avg_residual"], np.float32).T.copy()      # (2, 3): same pooled values
    cut, w, avg, codes = codec.compute_avg_residuals(k["nbits"], np.zeros((2, 1), np.float32), held)
    assert np.allclose(cut, k["expected_cutoffs"]) and np.allclose(w, k["expected_weights"])
    assert np.all(codes == 1) and np.isclose(avg, np.mean(held))


# ---------------------------------------------------------------------------------------------------
# codec pieces vs the oracle
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dim,nbits", [(128, 2), (128, 1), (128, 4), (64, 2), (24, 8), (128, 8)])
def test_decompress_bit_exact(oracle, dim, nbits):
    rng = np.random.default_rng(100 + dim + nbits)
    n, K = 1037, 57
    w = np.sort(rng.normal(0, 0.03, 1 << nbits).astype(np.float32))
    cent = (rng.normal(size=(dim, K)) / np.sqrt(dim)).astype(np.float32)
    codes = rng.integers(1, K + 1, size=n).astype(np.uint32)
    res = rng.integers(0, 256, size=(dim // 8 * nbits, n)).astype(np.uint8)
    assert_same_f32(codec.decompress(dim, nbits, cent, w, codes, res), oracle.decompress(dim, nbits, cent, w, codes, res),
                    f"decompress dim={dim} nbits={nbits}")


def test_normalize_bit_exact(oracle):
    rng = np.random.default_rng(101)
    for dim in (1, 3, 7, 128, 130):
        X = rng.normal(size=(dim, 77)).astype(np.float32)
        X[:, 5] = 0
        assert_same_f32(codec._normalize_array(X), oracle.normalize_array(X), f"normalize dim={dim}")


def test_maxsim_bit_exact(oracle):
    rng = np.random.default_rng(102)
    doclens = rng.integers(1, 12, size=150)
    Q = rng.normal(size=(128, 32)).astype(np.float32); D = rng.normal(size=(128, int(doclens.sum()))).astype(np.float32)
    pids = np.arange(1, 151)
    assert_same_f32(codec.maxsim(Q, D, pids, doclens), oracle.maxsim(Q, D, pids, doclens), "maxsim")


@pytest.mark.parametrize("dim,K,n", [(128, 300, 2100), (128, 1, 5), (128, 33, 31), (48, 20, 200), (128, 1000, 9001)])
def test_compress_bit_exact(oracle, dim, K, n):
    rng = np.random.default_rng(103 + K)
    cent = oracle.normalize_array(rng.normal(size=(dim, K)).astype(np.float32))
    embs = oracle.normalize_array(rng.normal(size=(dim, n)).astype(np.float32))
    embs[:, : min(K, n)] = cent[:, : min(K, n)]          # exact hits: residual bytes must be 0 there
    for nbits in (1, 2, 4):
        cut = np.sort(rng.normal(0, 0.05, (1 << nbits) - 1).astype(np.float32))
        codes, res = codec.compress(cent, cut, dim, nbits, embs)
        rc, rr = oracle.compress(cent, cut, dim, nbits, embs)
        assert np.array_equal(codes, rc) and np.array_equal(res, rr)
    assert np.array_equal(codec.compress_into_codes(cent, embs), oracle.compress_into_codes(cent, embs))


def test_compress_ties_pick_first_centroid(oracle):
    cent = np.zeros((128, 70), np.float32); cent[0, :] = 1.0       # all centroids identical
    embs = np.zeros((128, 40), np.float32); embs[0, :] = 0.5
    assert np.all(codec.compress_into_codes(cent, embs) == 1)
    cent[0, 37] = 2.0
    assert np.all(codec.compress_into_codes(cent, embs) == 38)


def test_nearest_centroid_mass_ties_take_the_list_path(oracle):
    """Nearly degenerate data (what a random-weight encoder produces): hundreds of identical or almost identical centroids,
    so that a point's candidate lists overflow and it is re-scored against ALL centroids (nearest_centroid_mfma_list_kernel)
    -- k-means assignments, centroids and compress codes still equal the oracle's, first index on ties."""
    rng = np.random.default_rng(77)
    base = oracle.normalize_array(rng.normal(size=(128, 6)).astype(np.float32))
    data = base[:, rng.integers(0, 6, size=5000)] + 1e-6 * rng.normal(size=(128, 5000)).astype(np.float32)
    data = np.asfortranarray(oracle.normalize_array(data.astype(np.float32)))
    K = 400
    init = np.asfortranarray(data[:, rng.permutation(5000)[:K]])          # ~67 near-copies of each of the 6 directions
    c, a, it = codec.kmeans(data, init, max_iters=3)
    rc, ra, rit = oracle.kmeans(data, init, max_iters=3)
    assert it == rit and np.array_equal(a, ra) and np.array_equal(c.view(np.uint32), rc.view(np.uint32))
    assert np.array_equal(codec.compress_into_codes(init, data), oracle.compress_into_codes(init, data))
    dup = np.asfortranarray(np.repeat(base, 50, axis=1))                      # 300 centroids, 50 exact copies of each
    assert np.array_equal(codec.compress_into_codes(dup, data), oracle.compress_into_codes(dup, data))


@pytest.mark.parametrize("products", ["1", "1-registers", "3"])
@pytest.mark.parametrize("scale", [1.0, 50.0, 2.0e-3, 1.0e3])
def test_nearest_centroid_near_ties_below_the_fp16_product_error(oracle, monkeypatch, products, scale):
    """The build's group lists come from ONE fp16 product per fp32 product (nearest_top_f16_dma_kernel, tiles by LDS-DMA;
    COLBERT_NEAREST_STAGING=registers: nearest_top_f16_kernel, its first form; COLBERT_NEAREST_PRODUCTS=3: the bf16 split).  Points that sit between two centroids with a score gap (1e-7 .. 1e-4) far below that product's error
    (~4e-4) must still get the oracle's code -- the refine margin carries the measured conversion errors -- in both modes
    (argmax dot, k-means distance), with long, short and (x 1e3 + one component beyond the fp16 range: three products) centroids."""
    monkeypatch.setenv("COLBERT_NEAREST_PRODUCTS", products[0])
    if products.endswith("registers"):
        if b"tuning build" not in clb.lib().clb_version():
            pytest.skip("the register-staged list kernel is a comparison kernel: tuning builds of the library only")
        monkeypatch.setenv("COLBERT_NEAREST_STAGING", "registers")
    rng = np.random.default_rng(211)
    K, n = 512, 6000
    cent = oracle.normalize_array(rng.normal(size=(128, K)).astype(np.float32))
    cent[:, 100:140] = cent[:, 60:100] + np.float32(3e-4) * rng.normal(size=(128, 40)).astype(np.float32)   # close pairs
    a, b = rng.integers(0, K, n), rng.integers(0, K, n)
    gap = (10.0 ** rng.uniform(-7, -4, n)).astype(np.float32)
    pts = cent[:, a] * (1 + gap) + cent[:, b] + np.float32(1e-3) * rng.normal(size=(128, n)).astype(np.float32)
    pts = np.asfortranarray(oracle.normalize_array(pts.astype(np.float32)))
    cent = np.asfortranarray(cent * np.float32(scale))
    if scale == 1.0e3:
        cent[3, 11] = np.float32(7.0e4)
        pts[:, :50] *= np.float32(7.0e4)                          # points beyond the fp16 range as well
    assert np.array_equal(codec.compress_into_codes(cent, pts), oracle.compress_into_codes(cent, pts))
    init = np.asfortranarray(cent[:, :300] if scale != 1.0e3 else cent[:, 12:312])
    c, asg, it = codec.kmeans(pts, init, max_iters=3)
    rc, ra, rit = oracle.kmeans(pts, init, max_iters=3)
    assert it == rit and np.array_equal(asg, ra)
    assert_same_f32(c, rc, "kmeans centroids")


@pytest.mark.parametrize("dim,n,K,bsize", [(128, 3000, 40, 1000), (128, 700, 64, 100), (16, 500, 9, 1000),
                                             (128, 6000, 200, 1000)])
def test_kmeans_bit_exact(oracle, dim, n, K, bsize):
    rng = np.random.default_rng(104 + K)
    data = oracle.normalize_array(rng.normal(size=(dim, n)).astype(np.float32))
    init = data[:, rng.permutation(n)[:K]]
    c, a, it = codec.kmeans(data, init, max_iters=6, point_bsize=bsize)
    rc, ra, rit = oracle.kmeans(data, init, max_iters=6, point_bsize=bsize)
    assert it == rit
    assert np.array_equal(a, ra)
    assert_same_f32(c, rc, "kmeans centroids")


def test_kmeans_fixed_point():
    rng = np.random.default_rng(105)                               # test/utils.jl:138-145
    data = rng.random((128, 90)).astype(np.float32)
    c, ids, _ = codec.kmeans(data, data[:, rng.permutation(90)], max_iters=10)
    assert np.array_equal(c[:, ids - 1], data)


def test_codec_stats_match_oracle(oracle):
    rng = np.random.default_rng(106)
    cent = oracle.normalize_array(rng.normal(size=(128, 50)).astype(np.float32))
    held = oracle.normalize_array(rng.normal(size=(128, 700)).astype(np.float32))
    for nbits in (1, 2, 4):
        cut, w, avg, codes = codec.compute_avg_residuals(nbits, cent, held)
        rcut, rw, ravg, rcodes = oracle.compute_avg_residuals(nbits, cent, held)
        assert np.array_equal(codes, rcodes)
        assert_same_f32(cut, rcut, "cutoffs"); assert_same_f32(w, rw, "weights")
        assert np.isclose(avg, ravg, rtol=1e-6)


@pytest.mark.parametrize("seed", range(int(os.environ.get("COLBERT_TEST_FUZZ_SEEDS", "8"))))
def test_index_build_stages_random_configurations(oracle, seed):
    """The index-build stages chained at randomly drawn shapes (dim a multiple of 8, nbits, K, sample and chunk sizes,
    k-means batch size): every stage's output is bit-identical to the oracle's, and each stage is fed the PRODUCT's
    output of the previous one, so a difference cannot hide behind a later stage."""
    rng = np.random.default_rng(8800 + seed)
    dim = int(rng.choice([8, 16, 24, 64, 128, 128, 128, 256]))
    nbits = int(rng.choice([1, 2, 2, 4, 8]))
    K = int(rng.integers(1, 400))
    n = int(rng.integers(K, K + 5000))
    centres = rng.normal(size=(dim, max(2, K // 3))).astype(np.float32)
    data = oracle.normalize_array((centres[:, rng.integers(0, centres.shape[1], size=n)]
                                   + 0.4 * rng.normal(size=(dim, n))).astype(np.float32))
    init = data[:, rng.permutation(n)[:K]]
    bsize = int(rng.choice([64, 1000, 4096]))
    c, a, it = codec.kmeans(data, init, max_iters=4, point_bsize=bsize)
    rc, ra, rit = oracle.kmeans(data, init, max_iters=4, point_bsize=bsize)
    assert it == rit and np.array_equal(a, ra)
    assert_same_f32(c, rc, "kmeans centroids")
    held = oracle.normalize_array(rng.normal(size=(dim, int(rng.integers(50, 900)))).astype(np.float32))
    cut, w, avg, hcodes = codec.compute_avg_residuals(nbits, c, held)
    rcut, rw, ravg, rhcodes = oracle.compute_avg_residuals(nbits, c, held)
    assert np.array_equal(hcodes, rhcodes)
    assert_same_f32(cut, rcut, "cutoffs"); assert_same_f32(w, rw, "weights")
    assert np.isclose(avg, ravg, rtol=1e-6)
    embs = oracle.normalize_array(rng.normal(size=(dim, int(rng.integers(1, 3000)))).astype(np.float32))
    codes, res = codec.compress(c, cut, dim, nbits, embs)
    rcodes, rres = oracle.compress(c, cut, dim, nbits, embs)
    assert np.array_equal(codes, rcodes) and np.array_equal(res, rres)
    assert_same_f32(codec.decompress(dim, nbits, c, w, codes, res), oracle.decompress(dim, nbits, c, w, codes, res), "decompress")
    ivf, lens = codec.build_ivf(codes, K)
    rivf, rlens = oracle.build_ivf(codes, K)
    assert np.array_equal(ivf, rivf) and np.array_equal(lens, rlens)


def test_build_ivf_matches_oracle(oracle):
    rng = np.random.default_rng(107)
    codes = rng.integers(1, 5001, size=200_000).astype(np.uint32)
    ivf, lens = codec.build_ivf(codes, 5000)
    rivf, rlens = oracle.build_ivf(codes, 5000)
    assert np.array_equal(ivf, rivf) and np.array_equal(lens, rlens)
    with pytest.raises(clb.BoundsError):
        codec.build_ivf(np.array([1, 7], np.uint32), 5)


def test_encoder_epilogue_bit_exact(oracle):
    rng = np.random.default_rng(108)
    dim, L, N = 128, 37, 9
    D = rng.normal(size=(dim, L, N)).astype(np.float32)
    ids = rng.integers(1, 40, size=(L, N)).astype(np.int32)
    skip = [1, 3, 7, 11, 999]
    out, doclens = codec.doc_epilogue(D, ids, skip)
    rout, rdl = oracle.doc_epilogue(D, ids, skip)
    assert np.array_equal(doclens, rdl)
    assert_same_f32(out, rout, "doc epilogue")
    assert_same_f32(codec.query_epilogue(D, ids, skip), oracle.query_epilogue(D, ids, skip), "query epilogue")


# ---------------------------------------------------------------------------------------------------
# search vs the oracle
# ---------------------------------------------------------------------------------------------------
def check_search(oracle, idx, Qs, k, nprobe=2, modes=(0, 1), pid_offset=0, wide=None):
    """Both search modes against the oracle; the two-pass mode with BOTH gather forms of pass 1 (mode 2 below = two-pass
    with the form the index statistics did not pick) and BOTH score-row formats (modes 3 / 4 = 8-bit cells with either gather
    form; the format of batches of 16+ queries, so their batch is the queries repeated up to 18)."""
    s = clb.Searcher(index=idx, pid_offset=pid_offset)
    if wide is not None:
        s.set_wide_select(wide)
    refs = [oracle.search(idx, Qs[:, :, j], nprobe=nprobe, k=k) for j in range(Qs.shape[2])]   # the oracle, once per query
    auto_form = s.pass1_gather[0]
    try:
        for mode in tuple(modes) + ((2, 3, 4) if 1 in modes else ()):
            if mode >= 1 and s.mode != 1:
                try:
                    s.set_mode(1)
                except clb.Unsupported:
                    continue
            s.set_mode(min(mode, 1))
            s.set_pass1_gather(-1 if mode not in (2, 4) else 1 - auto_form)
            if mode >= 3:
                try:
                    s.set_score_rows(1)
                except clb.Unsupported:                    # centroids of norm < 0.01
                    continue
                nq = Qs.shape[2]
                Qb = np.asfortranarray(np.concatenate([Qs] * (-(-18 // nq)), axis=2))
                bp, bs, bn = s.search_batch(Qb, k, nprobe=nprobe)
                for j in range(Qb.shape[2]):
                    rp, rs, rn = refs[j % nq]
                    assert np.array_equal(bp[:, j], rp + pid_offset) and bn[j] == rn, (mode, j)
                    assert_same_f32(bs[:, j], rs, f"batch scores mode={mode} q={j}")
                s.set_score_rows(0)
                continue
            s.set_score_rows(0)
            for j in range(Qs.shape[2]):
                rp, rs, rn = refs[j]
                pids, scores = s.search_embeddings(Qs[:, :, j], k, nprobe=nprobe)
                assert s.last_num_candidates == rn
                assert np.array_equal(pids, rp + pid_offset), (mode, j, np.nonzero(pids != rp + pid_offset)[0][:5])
                assert_same_f32(scores, rs, f"scores mode={mode} q={j}")
            # the batch entry point returns the same thing
            bp, bs, bn = s.search_batch(Qs, k, nprobe=nprobe)
            for j in range(Qs.shape[2]):
                rp, rs, rn = refs[j]
                assert np.array_equal(bp[:, j], rp + pid_offset) and bn[j] == rn
                assert_same_f32(bs[:, j], rs, f"batch scores mode={mode} q={j}")
    finally:
        s.close()


def test_search_small(oracle):
    idx = synthetic.make_index(seed=1, n_docs=300, K=64)
    check_search(oracle, idx, synthetic.make_queries(idx, 2, 3), k=10)


def test_search_medium(oracle):
    idx = synthetic.make_index(seed=3, n_docs=20_000, K=2048)
    check_search(oracle, idx, synthetic.make_queries(idx, 4, 4), k=1000)


def test_search_uniform_codes_and_nprobe(oracle):
    idx = synthetic.make_index(seed=5, n_docs=5000, K=512, topical=False)
    Qs = synthetic.make_queries(idx, 6, 2)
    check_search(oracle, idx, Qs, k=100, nprobe=1)
    check_search(oracle, idx, Qs, k=100, nprobe=4)
    check_search(oracle, idx, Qs, k=100, nprobe=9)


@pytest.mark.parametrize("nbits", [1, 4])
def test_search_other_nbits(oracle, nbits):
    idx = synthetic.make_index(seed=7 + nbits, n_docs=1500, K=128, nbits=nbits)
    check_search(oracle, idx, synthetic.make_queries(idx, 8, 2), k=50)


@pytest.mark.parametrize("T", [1, 5, 31, 40, 64, 100])
def test_search_query_lengths(oracle, T):
    idx = synthetic.make_index(seed=11, n_docs=1500, K=128)
    check_search(oracle, idx, synthetic.make_queries(idx, 12, 2, T=T), k=25)


@pytest.mark.parametrize("T", [32, 20])
def test_search_batch_sizes(oracle, T):
    """Batches of 8-15 queries take the shared-tile centroid kernel (8 queries per work-group, ragged last group), 16+
    the two-team kernel when it also writes the score table (mode 1: 16 queries per work-group, ragged last group and
    whole duplicate waves), smaller ones the per-query kernel.  All must give the oracle's result in both modes."""
    idx = synthetic.make_index(seed=19, n_docs=4000, K=1024)
    Qs = synthetic.make_queries(idx, 20, 35, T=T)
    ref = [oracle.search(idx, Qs[:, :, j], nprobe=2, k=50) for j in range(Qs.shape[2])]
    s = clb.Searcher(index=idx)
    try:
        for mode in (0, 1):
            s.set_mode(mode)
            for B in (7, 8, 11, 16, 19, 33, 35):
                bp, bs, bn = s.search_batch(np.asfortranarray(Qs[:, :, :B]), 50, nprobe=2)
                for j in range(B):
                    rp, rs, rn = ref[j]
                    assert np.array_equal(bp[:, j], rp) and bn[j] == rn, (mode, B, j)
                    assert_same_f32(bs[:, j], rs, f"mode={mode} B={B} q={j}")
    finally:
        s.close()


@pytest.mark.parametrize("scale", [1.0, 60.0, 3.0e-3, 1.0e3])
def test_centroid_products_of_large_batches(oracle, scale):
    """Batches of 16+ queries (two-pass mode) can build pass 1's fp16 score table from ONE fp16 product per fp32 product
    (clb_searcher_set_centroid_products(1)), the error bound carrying the measured fp16 conversion errors of queries and
    centroids; 3 (default) = the bf16 split every smaller batch uses.  Both give the oracle's result bit for bit -- also with centroids
    whose fp16 images are coarse (x 60: ulp 0.03), denormal-ish (x 3e-3) or out of range (x 1e3 -> three products)."""
    idx = synthetic.make_index(seed=41, n_docs=6000, K=2048)
    idx = dict(idx)
    idx["centroids"] = np.asfortranarray(idx["centroids"] * np.float32(scale))
    if scale == 1.0e3:
        idx["centroids"][5, 7] = np.float32(7.0e4)                    # beyond fp16's largest finite value
    Qs = synthetic.make_queries(idx, 42, 32)
    ref = [oracle.search(idx, Qs[:, :, j], nprobe=2, k=100) for j in range(Qs.shape[2])]
    s = clb.Searcher(index=idx)
    try:
        n_default, dc = s.centroid_products
        assert n_default == 3 and (dc > 0) == (scale < 1.0e3), (n_default, dc)
        n_one = 1 if dc > 0 else 3
        if scale < 1.0e3:       # the measured error is at most half an fp16 ulp per component of the longest centroid
            assert dc <= 2.0 ** -11 * float(np.linalg.norm(idx["centroids"], axis=0).max()) * 1.001 + 128 ** 0.5 * 2.0 ** -25
        for n in (1, 3, -1):
            s.set_centroid_products(n)
            assert s.centroid_products[0] == (n_one if n == 1 else 3)
            for B in (16, 32):
                bp, bs, bn = s.search_batch(np.asfortranarray(Qs[:, :, :B]), 100, nprobe=2)
                for j in range(B):
                    rp, rs, rn = ref[j]
                    assert np.array_equal(bp[:, j], rp) and bn[j] == rn, (n, B, j)
                    assert_same_f32(bs[:, j], rs, f"products={n} B={B} q={j}")
        with pytest.raises(clb.ArgumentError):
            s.set_centroid_products(2)
    finally:
        s.close()


def test_search_mixed_shapes_on_one_handle(oracle):
    """One Searcher, calls of different shapes in the order that used to leave the tuned-path buffers unallocated or
    short (round-2 advisor finding: a T > 128 query switched `general` on for every later workspace growth): T = 150
    first (general path), then T = 64 with nprobe = 4 and a batch of 32, then T = 32 / nprobe 2, then T = 150 again."""
    idx = synthetic.make_index(seed=61, n_docs=3000, K=256, doclen_mean=24, doclen_std=6)
    s = clb.Searcher(index=idx)
    try:
        def run(T, nprobe, B, k, seed):
            Qs = synthetic.make_queries(idx, seed, B, T=T)
            if B == 1:
                got = [s.search_embeddings(Qs[:, :, 0], k, nprobe=nprobe)]
            else:
                bp, bs, _ = s.search_batch(Qs, k, nprobe=nprobe)
                got = [(bp[:, j], bs[:, j]) for j in range(B)]
            for j, (pids, scores) in enumerate(got):
                rp, rs, _ = oracle.search(idx, Qs[:, :, j], nprobe=nprobe, k=k)
                assert np.array_equal(pids, rp), (T, nprobe, B, j)
                assert_same_f32(scores, rs, f"T={T} nprobe={nprobe} B={B} q={j}")
        for mode in (0, 1):
            s.set_mode(mode)
            run(64, 2, 1, 20, 70)
            run(150, 2, 1, 20, 71)
            run(64, 4, 32, 20, 72)
            run(32, 2, 9, 20, 73)
            run(150, 3, 2, 20, 74)
            run(100, 9, 5, 20, 75)
    finally:
        s.close()


def test_search_ragged_and_empty_passages(oracle):
    idx = synthetic.make_index(seed=13, n_docs=800, K=64, doclen_mean=20, doclen_std=30)
    # force zero-length passages (the reference tolerates them: _build_emb2pid test 3)
    rng = np.random.default_rng(14)
    dl = idx["doclens"].copy()
    dl[rng.integers(0, 800, size=60)] = 0
    dl[0] = 0; dl[-1] = 0
    n_emb = int(dl.sum())
    idx2 = dict(idx, doclens=dl, codes=idx["codes"][:n_emb], residuals=np.asfortranarray(idx["residuals"][:, :n_emb]))
    idx2["ivf"], idx2["ivf_lengths"] = synthetic.build_ivf(idx2["codes"], 64)
    check_search(oracle, idx2, synthetic.make_queries(idx2, 15, 3), k=20)


def test_exact_steps_shared_by_consecutive_passages(oracle):
    """Round 6: the exact kernel starts a wave's next passage in the free row slots of a passage's last 16-row step.  A batch of
    32 queries with ~1 500 listed passages each gives every wave a dozen passages in a row; the lengths are ragged on purpose
    (1, 2, ... rows: several passages end inside one step and the second one is padded; 16 / 17 / 32 / 33: a boundary exactly at
    or next to a step edge; > 256: the identity mapping next to masked neighbours; empty passages in between)."""
    idx = synthetic.make_index(seed=31, n_docs=3000, K=64, doclen_mean=12, doclen_std=14, doclen_max=400)
    dl = idx["doclens"].copy()
    forced = [1, 2, 3, 15, 16, 17, 31, 32, 33, 47, 48, 49, 255, 256, 257, 300, 0, 1, 16, 1]
    rng = np.random.default_rng(32)
    for j, pid in enumerate(rng.choice(3000, size=20 * len(forced), replace=False)):
        dl[pid] = forced[j % len(forced)]
    n_emb = int(dl.sum())
    if n_emb > idx["codes"].shape[0]:                        # (the forced lengths may ask for more embeddings than were generated)
        extra = n_emb - idx["codes"].shape[0]
        idx["codes"] = np.concatenate([idx["codes"], idx["codes"][:extra]])
        idx["residuals"] = np.asfortranarray(np.concatenate([idx["residuals"], idx["residuals"][:, :extra]], axis=1))
    idx2 = dict(idx, doclens=dl, codes=idx["codes"][:n_emb], residuals=np.asfortranarray(idx["residuals"][:, :n_emb]))
    idx2["ivf"], idx2["ivf_lengths"] = synthetic.build_ivf(idx2["codes"], 64)
    Qs = synthetic.make_queries(idx2, 33, 32)
    fewest = min(oracle.search(idx2, Qs[:, :, j], nprobe=2, k=1)[2] for j in range(Qs.shape[2]))
    assert fewest > 600, fewest
    check_search(oracle, idx2, Qs, k=min(1500, fewest), modes=(1,))


def test_search_long_passages(oracle):
    """Passages longer than 256 embeddings do not fit the row mask of the two-pass mode and take every row in the
    exact pass; shorter ones in the same index use the mask.  Both must equal the oracle, in both modes."""
    idx = synthetic.make_index(seed=23, n_docs=400, K=128, doclen_mean=230, doclen_std=90, doclen_max=600)
    assert int(idx["doclens"].max()) > 256 and int(idx["doclens"].min()) < 128
    check_search(oracle, idx, synthetic.make_queries(idx, 24, 3), k=50)
    check_search(oracle, idx, synthetic.make_queries(idx, 25, 9, T=20), k=30, modes=(1,))


def test_search_many_candidates(oracle):
    """More than 32 768 candidates per query: the selection kernels leave their register-cached path (strided
    re-reads, block-wise compaction).  Few, huge IVF lists make almost every passage a candidate."""
    idx = synthetic.make_index(seed=27, n_docs=45_000, K=32, doclen_mean=40, doclen_std=8)
    Qs = synthetic.make_queries(idx, 28, 2)
    s = clb.Searcher(index=idx)
    s.search_embeddings(Qs[:, :, 0], 10, nprobe=2)
    assert s.last_num_candidates > 32_768 + 1024
    s.close()
    check_search(oracle, idx, Qs, k=1000)


@pytest.mark.parametrize("seed", range(int(os.environ.get("COLBERT_TEST_FUZZ_SEEDS", "10"))))   # more seeds for a one-off sweep
def test_search_random_configurations(oracle, seed):
    """Randomly drawn shapes (corpus size, centroid count, nbits, passage lengths, query length, batch, k, nprobe,
    topical or uniform codes, scaled queries): pids identical and scores bit-identical to the oracle, both modes."""
    rng = np.random.default_rng(7000 + seed)
    n_docs = int(rng.integers(40, 6000))
    K = int(2 ** rng.integers(3, 11))
    nbits = int(rng.choice([1, 2, 2, 2, 4]))
    mean = float(rng.integers(4, 120))
    idx = synthetic.make_index(seed=7100 + seed, n_docs=n_docs, K=K, nbits=nbits, doclen_mean=mean,
                               doclen_std=float(rng.integers(0, 40)), topical=bool(rng.integers(0, 2)))
    T = int(rng.choice([1, 3, 17, 32, 32, 32, 33, 48]))
    Qs = synthetic.make_queries(idx, 7200 + seed, int(rng.integers(1, 10)), T=T)
    if rng.integers(0, 3) == 0:
        Qs = np.asfortranarray(Qs * np.float32(rng.choice([0.25, 3.0, 40.0])))   # the reference does not require unit queries
    k = int(min(n_docs, rng.choice([1, 10, 100, 1000, n_docs])))
    nprobe = int(min(K, rng.choice([1, 2, 2, 3, 8])))
    # the reference raises BoundsError when k exceeds a query's candidate count (searching.jl:127): so must the product;
    # the comparison itself then runs at the largest k every query of the batch can fill
    counts = [oracle.search(idx, Qs[:, :, j], nprobe=nprobe, k=1)[2] for j in range(Qs.shape[2])]
    fewest = min(counts)
    if k > fewest:
        j = int(np.argmin(counts))
        srch = clb.Searcher(index=idx)
        try:
            for mode in (0, 1):
                try:
                    srch.set_mode(mode)
                except clb.Unsupported:
                    continue
                with pytest.raises(clb.BoundsError):
                    srch.search_embeddings(Qs[:, :, j], k, nprobe=nprobe)
        finally:
            srch.close()
        k = fewest
    check_search(oracle, idx, Qs, k=k, nprobe=nprobe)


@pytest.mark.parametrize("case", ["many", "few", "ties", "scaled", "uniform"])
def test_wide_selection_matches_oracle(oracle, case):
    """The selection step with sixteen work-groups per query and one launch per radix pass (what a 10 M-passage shard
    takes by default) forced on small indexes: more candidates than one work-group keeps in registers, fewer candidates
    than k, tied scores, un-normalised queries (the unsafe path lists everything), uniform codes."""
    if case == "many":
        idx = synthetic.make_index(seed=27, n_docs=45_000, K=32, doclen_mean=40, doclen_std=8)
        Qs, k = synthetic.make_queries(idx, 28, 2), 1000
    elif case == "few":
        idx = synthetic.make_index(seed=1, n_docs=300, K=64)
        Qs = synthetic.make_queries(idx, 2, 5)
        k = min(oracle.search(idx, Qs[:, :, j], nprobe=2, k=1)[2] for j in range(5))      # k = the fewest candidates: n <= k
    elif case == "ties":
        idx = synthetic.make_index(seed=17, n_docs=400, K=32, constant_doclen=True, doclen_mean=16)
        for p in range(0, 400, 2):
            idx["codes"][(p + 1) * 16:(p + 2) * 16] = idx["codes"][p * 16:(p + 1) * 16]
            idx["residuals"][:, (p + 1) * 16:(p + 2) * 16] = idx["residuals"][:, p * 16:(p + 1) * 16]
        idx["ivf"], idx["ivf_lengths"] = synthetic.build_ivf(idx["codes"], 32)
        Qs, k = synthetic.make_queries(idx, 18, 9), 60
    elif case == "scaled":
        idx = synthetic.make_index(seed=3, n_docs=6000, K=512)
        Qs = synthetic.make_queries(idx, 4, 8)
        Qs = np.asfortranarray(Qs * np.float32(1.0e5))                                   # fp16 table overflows: unsafe query
        k = 200
    else:
        idx = synthetic.make_index(seed=5, n_docs=5000, K=512, topical=False)
        Qs, k = synthetic.make_queries(idx, 6, 11), 100
    check_search(oracle, idx, Qs, k=k, modes=(1,), wide=1)


def test_search_ties_keep_ascending_pid(oracle):
    """Duplicate passages score identically; the stable sortperm keeps the lower pid first."""
    idx = synthetic.make_index(seed=17, n_docs=400, K=32, constant_doclen=True, doclen_mean=16)
    L = 16
    for p in range(0, 400, 2):                                   # passage p+1 := copy of passage p
        idx["codes"][(p + 1) * L:(p + 2) * L] = idx["codes"][p * L:(p + 1) * L]
        idx["residuals"][:, (p + 1) * L:(p + 2) * L] = idx["residuals"][:, p * L:(p + 1) * L]
    idx["ivf"], idx["ivf_lengths"] = synthetic.build_ivf(idx["codes"], 32)
    check_search(oracle, idx, synthetic.make_queries(idx, 18, 2), k=60)


def test_search_tied_centroids(oracle):
    """Many identical centroids: every score ties, the bf16x3 candidate lists overflow and the refine kernel must
    fall back to the exhaustive canonical scan -- the selected centroids are still the lowest indices
    (partialsortperm's tie rule), so candidates and results equal the oracle's."""
    idx = synthetic.make_index(seed=29, n_docs=600, K=96)
    C = idx["centroids"].copy(order="F")
    C[:, 10:70] = C[:, [10]]                                       # 60 identical centroids
    C[:, 80:90] = C[:, [10]] * np.float32(1.0)                     # and 10 more copies further up
    idx["centroids"] = C
    Qs = synthetic.make_queries(idx, 30, 2)
    Qs[:, :8, 0] = (C[:, [10]] / np.linalg.norm(C[:, 10])).astype(np.float32)   # tokens aligned with the tied group
    check_search(oracle, idx, Qs, k=40)
    s = clb.Searcher(index=idx)
    for j in range(2):
        assert np.array_equal(s.retrieve(Qs[:, :, j]),
                              oracle.retrieve(idx["ivf"], idx["ivf_lengths"], idx["centroids"],
                                              oracle.build_emb2pid(idx["doclens"]), 2, Qs[:, :, j]))
    s.close()


def test_device_merge_kernel_matches_unsharded(oracle):
    """The multi-GPU data path on one GPU: per-shard device search -> stacked (world, B, k) records (what
    all_gather_into_tensor produces) -> clb_merge_topk_device == unsharded oracle result."""
    torch = pytest.importorskip("torch")
    from colbert_jl_amd.distributed import DeviceSearch, merge_gathered, merge_packed
    from colbert_jl_amd.sharding import shard_index
    idx = synthetic.make_index(seed=33, n_docs=5000, K=512)
    Qs = synthetic.make_queries(idx, 34, 3)                       # B*k odd-sized blocks exercise the padding
    k, world = 301, 4
    Qdev = torch.from_numpy(np.ascontiguousarray(Qs.transpose(2, 1, 0))).cuda()
    gp, gs, packed, keep = [], [], [], []
    for rnk in range(world):
        sub, off = shard_index(idx, rnk, world)
        s = clb.Searcher(index=sub, pid_offset=off)
        run = DeviceSearch(s, 32, Qs.shape[2], k, 2)
        if rnk == 0:        # the C ABI takes a bare pointer: a short, strided or mistyped tensor must never reach a kernel
            for bad in (Qdev[:2], Qdev[:, :, ::2], Qdev.double(), Qdev.cpu()):
                with pytest.raises(clb.ArgumentError):
                    run(bad)
                with pytest.raises(clb.ArgumentError):
                    run.phase1(bad)
        p, sc = run(Qdev)
        torch.cuda.synchronize()
        gp.append(p.clone()); gs.append(sc.clone()); packed.append(run.packed.clone()); keep.append(s)
    mp, ms = merge_gathered(torch.stack(gp), torch.stack(gs), k)
    # the packed layout one all-gather produces: (world, packed_topk_bytes) -> clb_merge_topk_packed_device
    pp, ps = merge_packed(torch.stack(packed), Qs.shape[2], k)
    torch.cuda.synchronize()
    assert torch.equal(mp, pp) and torch.equal(ms, ps)
    mp = mp.cpu().numpy(); ms = ms.cpu().numpy()
    for j in range(Qs.shape[2]):
        rp, rs, _ = oracle.search(idx, Qs[:, :, j], 2, k)
        assert np.array_equal(mp[j], rp)
        assert_same_f32(ms[j], rs, "device-merged scores")
    for s in keep:
        s.close()


def test_bench_sharding_path_matches_unsharded(oracle):
    """The exact data path of `bench.py --gpus N` on one GPU: shards generated per block range, device search with
    packed outputs, the packed blocks stacked as an all-gather would, merged -- equal to the oracle on the full index."""
    torch = pytest.importorskip("torch")
    from colbert_jl_amd.distributed import DeviceSearch, merge_packed
    full = synthetic.make_index(seed=2024, n_docs=6000, K=512, n_blocks=8)
    Qs = synthetic.make_topic_queries(full["centroids"], seed=77, n_queries=9)
    Qdev = torch.from_numpy(np.ascontiguousarray(Qs.transpose(2, 1, 0))).cuda()
    k, world = 200, 4
    packed, keep = [], []
    for rank in range(world):
        sh = synthetic.make_index(seed=2024, n_docs=6000, K=512, n_blocks=8, blocks=range(2 * rank, 2 * rank + 2))
        s = clb.Searcher(index=sh, pid_offset=int(sh["pid_offset"]))
        run = DeviceSearch(s, 32, 9, k, 2)
        run(Qdev)
        torch.cuda.synchronize()
        packed.append(run.packed.clone()); keep.append(s)
    mp, ms = merge_packed(torch.stack(packed), 9, k)
    torch.cuda.synchronize()
    mp = mp.cpu().numpy(); ms = ms.cpu().numpy()
    for j in range(9):
        rp, rs, _ = oracle.search(full, Qs[:, :, j], 2, k)
        assert np.array_equal(mp[j], rp)
        assert_same_f32(ms[j], rs, "bench sharding path")
    for s in keep:
        s.close()


@pytest.mark.parametrize("world,k,wide,nq,rows", [(4, 200, -1, 9, 0), (8, 1000, -1, 19, 0), (2, 5000, -1, 9, 0), (4, 200, 1, 19, 0), (2, 5000, 1, 9, 0),
                                                  (2, 300, -1, 32, 0), (8, 1000, -1, 19, 1), (2, 300, 1, 32, 1)])
def test_two_phase_sharded_search(oracle, world, k, wide, nq, rows):
    """clb_search_shard_phase1/2 on `world` shards of one index (the all-gather is simulated by stacking the shards'
    score blocks): every shard cuts at the global k-th approximate score, the merged result equals the oracle's on
    the full index, and the shards together list far fewer passages than with shard-local thresholds.  k = 5000
    exceeds what some queries can return (padding, tau = -inf).  Batches of 16+ queries on shards that share a bound build
    their score table from one fp16 product (clb_searcher_set_centroid_products' default on a shard group): nq = 19.
    rows = 1 (round 6): the same with the score tables as 8-bit rows on every shard (clb_searcher_set_score_rows, set alike on all
    shards: the global threshold is cut with a bound that covers every shard's table)."""
    torch = pytest.importorskip("torch")
    from colbert_jl_amd.distributed import DeviceSearch, merge_packed, share_bound_consts
    full = synthetic.make_index(seed=2024, n_docs=8000, K=512, n_blocks=8)
    Qs = synthetic.make_topic_queries(full["centroids"], seed=78, n_queries=nq)
    Qdev = torch.from_numpy(np.ascontiguousarray(Qs.transpose(2, 1, 0))).cuda()
    per = 8 // world
    kk = min(k, 4096)
    runs, keep = [], []
    for rank in range(world):
        sh = synthetic.make_index(seed=2024, n_docs=8000, K=512, n_blocks=8, blocks=range(per * rank, per * rank + per))
        s = clb.Searcher(index=sh, pid_offset=int(sh["pid_offset"]))
        s.set_wide_select(wide)                                          # 1: the sixteen-work-group selection in both phases
        s.set_score_rows(rows)
        runs.append(DeviceSearch(s, 32, nq, kk, 2)); keep.append(s)
    tops = torch.stack([r.phase1(Qdev).clone() for r in runs])           # (world, B, k) as an all-gather delivers
    torch.cuda.synchronize()
    # the protocol checks itself: phase 2 against other shards' scores is refused until the shards share ONE error bound
    with pytest.raises(clb.ArgumentError):
        runs[0].phase2(Qdev, tops)
    share_bound_consts(keep)
    assert all(s.centroid_products[0] == 1 for s in keep)                 # a shard group: single-product tables for 16+ queries
    runs[0].phase1(Qdev)                                                  # (the refused call consumed nothing; redo phase 1)
    packed = []
    for r in runs:
        r.phase2(Qdev, tops)
        torch.cuda.synchronize()
        packed.append(r.packed.clone())
    mp, ms = merge_packed(torch.stack(packed), nq, kk)
    torch.cuda.synchronize()
    mp = mp.cpu().numpy(); ms = ms.cpu().numpy()
    for j in range(nq):
        n_all = oracle.retrieve(full["ivf"], full["ivf_lengths"], full["centroids"],
                                oracle.build_emb2pid(full["doclens"]), 2, Qs[:, :, j]).size
        ke = min(kk, n_all)
        rp, rs, _ = oracle.search(full, Qs[:, :, j], 2, ke)
        assert np.array_equal(mp[j, :ke], rp), (world, j)
        assert_same_f32(ms[j, :ke], rs, "two-phase sharded search")
        assert np.all(mp[j, ke:] == 0)
    for s in keep:
        s.close()


@pytest.mark.parametrize("shuffled", [False, True])
def test_candidate_marking_large_shard(oracle, shuffled):
    """More than 16 bitmap slices (2.2 M passages): a batch marks its candidates slice by slice, every slice cutting its part
    out of each IVF list by binary search -- valid because the lists hold non-decreasing passage ids.  With the entries of
    every list shuffled (the reference never needs an order inside a list) the handle must notice at load and keep the
    atomic path.  Candidates and results equal the oracle's either way."""
    idx = synthetic.make_index(seed=61, n_docs=2_200_000, K=4096, doclen_mean=3, doclen_std=1)
    if shuffled:
        rng = np.random.default_rng(62)
        ivf = idx["ivf"].copy()
        off = np.concatenate([[0], np.cumsum(idx["ivf_lengths"])])
        for c in rng.integers(0, 4096, size=600):                  # enough lists to hit the probed ones
            ivf[off[c]:off[c + 1]] = rng.permutation(ivf[off[c]:off[c + 1]])
        top = np.argsort(idx["ivf_lengths"])[-64:]
        for c in top:
            ivf[off[c]:off[c + 1]] = rng.permutation(ivf[off[c]:off[c + 1]])
        idx = dict(idx, ivf=ivf)
    Qs = synthetic.make_queries(idx, 63, 8)                        # 8: the smallest batch that marks slice by slice
    s = clb.Searcher(index=idx)
    try:
        bp, bs, bn = s.search_batch(Qs, 100, nprobe=2)
        for j in (0, 3, 7):
            rp, rs, rn = oracle.search(idx, Qs[:, :, j], nprobe=2, k=100)
            assert bn[j] == rn and np.array_equal(bp[:, j], rp), j
            assert_same_f32(bs[:, j], rs, f"large shard q={j}")
        cand = s.retrieve(Qs[:, :, 0])
        assert np.array_equal(cand, oracle.retrieve(idx["ivf"], idx["ivf_lengths"], idx["centroids"],
                                                    oracle.build_emb2pid(idx["doclens"]), 2, Qs[:, :, 0]))
    finally:
        s.close()


@pytest.mark.parametrize("n_docs", [131071, 131073, 262145])
def test_candidate_bitmap_slice_boundaries(oracle, n_docs):
    """Batches mark their candidates in LDS slices of 131 072 passages (mark_count_kernel): corpora that end one passage
    before / after a slice boundary, a batch of 9 (sliced path) and single queries (atomic path) against the oracle."""
    idx = synthetic.make_index(seed=n_docs, n_docs=n_docs, K=2048, doclen_mean=12.0, doclen_std=3.0)
    Qs = synthetic.make_topic_queries(idx["centroids"], seed=81, n_queries=9)
    s = clb.Searcher(index=idx)
    k = 50
    bp, bs, _ = s.search_batch(Qs, k, nprobe=2)
    for j in range(9):
        rp, rs, _ = oracle.search(idx, Qs[:, :, j], 2, k)
        assert np.array_equal(bp[:, j], rp), j
        assert_same_f32(bs[:, j], rs, "sliced candidate marking")
        if j < 2:
            p1, s1 = s.search_embeddings(Qs[:, :, j], k=k)
            assert np.array_equal(p1, rp)
            cand = s.retrieve(Qs[:, :, j])
            assert np.array_equal(cand, oracle.retrieve(idx["ivf"], idx["ivf_lengths"], idx["centroids"],
                                                        oracle.build_emb2pid(idx["doclens"]), 2, Qs[:, :, j]))
    s.close()


def test_captured_graph_replays_the_search(oracle):
    """DeviceSearch.capture: the launches of one search captured as a HIP graph over a static query buffer; every
    replay (new queries written into the buffer) gives the oracle's result."""
    import torch
    from colbert_jl_amd.distributed import DeviceSearch
    idx = synthetic.make_index(seed=23, n_docs=6000, K=1024)
    k = 60
    s = clb.Searcher(index=idx)
    try:
        for B in (1, 17):
            Qs = synthetic.make_queries(idx, 31, 3 * B)
            Qdev = torch.from_numpy(np.ascontiguousarray(Qs.transpose(2, 1, 0))).cuda()
            run = DeviceSearch(s, 32, B, k, 2)
            q_static = Qdev[:B].clone()
            graph = run.capture(q_static)
            for it in range(3):
                q_static.copy_(Qdev[it * B:(it + 1) * B])
                graph.replay()
                torch.cuda.synchronize()
                for j in range(B):
                    rp, rs, rn = oracle.search(idx, Qs[:, :, it * B + j], nprobe=2, k=k)
                    assert np.array_equal(run.out_p[j].cpu().numpy(), rp), (B, it, j)
                    assert_same_f32(run.out_s[j].cpu().numpy(), rs, f"graph B={B} it={it} q={j}")
            del graph
    finally:
        s.close()


def test_batches_in_flight_equal_serial_results():
    """Forty batches through two workspace slots on two streams (what bench.py times) give exactly the results the same
    batches give one after the other on one stream."""
    torch = pytest.importorskip("torch")
    from colbert_jl_amd.distributed import DeviceSearch
    idx = synthetic.make_index(seed=2026, n_docs=20000, K=1024)
    nb, B, k = 40, 8, 200
    Qs = synthetic.make_topic_queries(idx["centroids"], seed=80, n_queries=nb * B)
    Qdev = torch.from_numpy(np.ascontiguousarray(Qs.transpose(2, 1, 0))).cuda()
    s = clb.Searcher(index=idx)
    runs = [DeviceSearch(s, 32, B, k, 2, slot=i) for i in range(2)]
    serial = []
    for i in range(nb):
        p, sc = runs[0](Qdev[i * B:(i + 1) * B])
        torch.cuda.synchronize()
        serial.append((p.clone(), sc.clone()))
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    got = [None] * nb
    for i in range(nb):
        with torch.cuda.stream(streams[i & 1]):
            p, sc = runs[i & 1](Qdev[i * B:(i + 1) * B])
            got[i] = (p.clone(), sc.clone())          # stream-ordered behind the search of batch i
    torch.cuda.synchronize()
    for i in range(nb):
        assert torch.equal(got[i][0], serial[i][0]) and torch.equal(got[i][1], serial[i][1]), i
    s.close()


def test_two_phase_batches_in_flight_use_their_own_slot(oracle):
    """Several batches of the two-phase search in flight on different streams (what bench.py does with more than one
    rank): each continues on its own workspace slot -- phase 1 of the next batch must not disturb phase 2 of the
    previous one -- and a phase 2 on the wrong slot is refused."""
    torch = pytest.importorskip("torch")
    from colbert_jl_amd.distributed import DeviceSearch
    idx = synthetic.make_index(seed=2025, n_docs=6000, K=256)
    Qs = synthetic.make_topic_queries(idx["centroids"], seed=79, n_queries=4 * 5)
    Qdev = torch.from_numpy(np.ascontiguousarray(Qs.transpose(2, 1, 0))).cuda()
    s = clb.Searcher(index=idx)
    k, B, nslots = 100, 5, 4
    runs = [DeviceSearch(s, 32, B, k, 2, slot=i) for i in range(nslots)]
    streams = [torch.cuda.Stream() for _ in range(nslots)]
    torch.cuda.synchronize()
    tops = []
    for i in range(nslots):                                   # all first halves, one per stream
        with torch.cuda.stream(streams[i]):
            tops.append(runs[i].phase1(Qdev[i * B:(i + 1) * B]).unsqueeze(0).contiguous())
    with pytest.raises(clb.ArgumentError):                    # batch 1's second half on batch 0's slot
        with torch.cuda.stream(streams[0]):
            runs[0].phase2(Qdev[B:2 * B], tops[1])
    for i in reversed(range(nslots)):                         # second halves in the opposite order
        with torch.cuda.stream(streams[i]):
            runs[i].phase2(Qdev[i * B:(i + 1) * B], tops[i])
    torch.cuda.synchronize()
    for i in range(nslots):
        p = runs[i].out_p.cpu().numpy(); sc = runs[i].out_s.cpu().numpy()
        for j in range(B):
            rp, rs, _ = oracle.search(idx, Qs[:, :, i * B + j], 2, k)
            assert np.array_equal(p[j], rp), (i, j)
            assert_same_f32(sc[j], rs, "two-phase search on slot %d" % i)
    s.close()


def test_library_rccl_communicator_single_rank(oracle):
    """clb_comm_* (the exchange step on RCCL inside the library, for hosts without torch.distributed) with one rank: the
    packed top-k all-gather + merge and the bound-constant all-reduce reproduce the unsharded result.  More ranks need
    more GPUs than this box has; the entry points are the same."""
    torch = pytest.importorskip("torch")
    from colbert_jl_amd.distributed import DeviceSearch, LibraryComm, merge_packed
    idx = synthetic.make_index(seed=41, n_docs=3000, K=256)
    Qs = synthetic.make_queries(idx, 42, 4)
    Qdev = torch.from_numpy(np.ascontiguousarray(Qs.transpose(2, 1, 0))).cuda()
    uid = LibraryComm.unique_id()
    assert len(uid) == 128
    comm = LibraryComm(0, 0, 1, uid)
    s = clb.Searcher(index=idx)
    before = s.bound_consts.copy()
    assert np.array_equal(comm.sync_bound_consts(s), before)
    k = 64
    run = DeviceSearch(s, 32, 4, k, 2)
    tops = comm.all_gather(run.phase1(Qdev))                     # (1, B, k)
    run.phase2(Qdev, tops)
    gathered = comm.all_gather(run.packed)                       # (1, packed bytes)
    mp, ms = merge_packed(gathered, 4, k)
    torch.cuda.synchronize()
    for j in range(4):
        rp, rs, _ = oracle.search(idx, Qs[:, :, j], 2, k)
        assert np.array_equal(mp[j].cpu().numpy(), rp)
        assert_same_f32(ms[j].cpu().numpy(), rs, "library communicator")
    x = torch.arange(6, dtype=torch.float32, device="cuda")
    assert torch.equal(comm.all_reduce_max_(x.clone()), x)
    with pytest.raises(clb.ArgumentError):
        LibraryComm(0, 1, 1, uid)                                # rank out of range
    comm.close(); s.close()


def test_search_bounds_error_and_padding(oracle):
    idx = synthetic.make_index(seed=19, n_docs=300, K=64)
    Q = synthetic.make_queries(idx, 20, 1)
    s = clb.Searcher(index=idx)
    cand = s.retrieve(Q[:, :, 0])
    assert np.array_equal(cand, oracle.retrieve(idx["ivf"], idx["ivf_lengths"], idx["centroids"],
                                                oracle.build_emb2pid(idx["doclens"]), 2, Q[:, :, 0]))
    with pytest.raises(clb.BoundsError):                         # searching.jl:127
        s.search_embeddings(Q[:, :, 0], k=cand.size + 1)
    pids, scores, n = s.search_batch(Q, cand.size + 5, pad_short=True)
    rp, rs, _ = oracle.search(idx, Q[:, :, 0], 2, cand.size)
    assert np.array_equal(pids[: cand.size, 0], rp) and np.all(pids[cand.size:, 0] == 0)
    assert np.all(np.isneginf(scores[cand.size:, 0]))
    with pytest.raises(clb.BoundsError):                         # partialsortperm(v, 1:nprobe) with nprobe > K
        s.search_embeddings(Q[:, :, 0], k=1, nprobe=65)
    s.close()


def test_batch_nprobe1_and_short_results(oracle):
    """A batch of 9 (shared-tile centroid kernel, 2-D row sweep) with nprobe = 1 and with k above the number of
    candidates of every query: padded tails, otherwise the oracle's result."""
    idx = synthetic.make_index(seed=21, n_docs=300, K=64)
    Qs = synthetic.make_queries(idx, 22, 9)
    s = clb.Searcher(index=idx)
    try:
        for mode in (0, 1):
            s.set_mode(mode)
            for nprobe in (1, 2):
                k = 280
                bp, bs, bn = s.search_batch(Qs, k, nprobe=nprobe, pad_short=True)
                for j in range(9):
                    n = int(bn[j])
                    rp, rs, rn = oracle.search(idx, Qs[:, :, j], nprobe, min(k, n))
                    assert rn == n
                    kk = min(k, n)
                    assert np.array_equal(bp[:kk, j], rp), (mode, nprobe, j)
                    assert_same_f32(bs[:kk, j], rs, f"mode={mode} nprobe={nprobe} q={j}")
                    assert np.all(bp[kk:, j] == 0) and np.all(np.isneginf(bs[kk:, j]))
    finally:
        s.close()


def test_create_rejects_bad_codes():
    idx = synthetic.make_index(seed=21, n_docs=50, K=16)
    bad = dict(idx); bad["codes"] = idx["codes"].copy(); bad["codes"][3] = 17
    with pytest.raises(clb.DomainError):
        clb.Searcher(index=bad)
    bad = dict(idx); bad["ivf"] = idx["ivf"].copy(); bad["ivf"][0] = idx["codes"].size + 1
    with pytest.raises(clb.BoundsError):
        clb.Searcher(index=bad)


def test_two_pass_error_bound_and_superset():
    """The approximate pass stays within its proven bound eps of the exact (canonical fp32) score, the
    observed error is far smaller than the bound, and {approx >= tau - 2 eps} contains the exact top-k."""
    idx = synthetic.make_index(seed=27, n_docs=30_000, K=4096)
    Qs = synthetic.make_queries(idx, 28, 4)
    s = clb.Searcher(index=idx)
    k = 500
    for j in range(Qs.shape[2]):
        d = s.debug_scores(Qs[:, :, j], k)
        err = np.abs(d["approx"].astype(np.float64) - d["exact"].astype(np.float64))
        assert d["eps"] > 0 and err.max() <= d["eps"], (err.max(), d["eps"])
        assert err.max() <= d["eps"] / 2, ("bound unexpectedly tight", err.max(), d["eps"])
        order = np.lexsort((d["pids"], -d["exact"].astype(np.float64)))[:k]
        selected = d["approx"] >= np.float32(d["tau"]) - np.float32(2) * np.float32(d["eps"])
        assert selected[order].all()
        assert selected.sum() == d["n_rescore"] and k <= d["n_rescore"] <= max(4 * k, 2000)
    s.close()


def test_sharded_search_merges_to_unsharded(oracle):
    """contiguous pid shards + merge of per-shard top-k == unsharded result."""
    idx = synthetic.make_index(seed=23, n_docs=6000, K=512)
    Qs = synthetic.make_queries(idx, 24, 3)
    k = 200
    from colbert_jl_amd.sharding import merge_topk_host, shard_index
    parts = [shard_index(idx, r, 4) for r in range(4)]
    res = []
    for sub, off in parts:
        s = clb.Searcher(index=sub, pid_offset=off)
        res.append(s.search_batch(Qs, k, pad_short=True)[:2])
        s.close()
    for j in range(Qs.shape[2]):
        rp, rs, _ = oracle.search(idx, Qs[:, :, j], 2, k)
        mp, ms = merge_topk_host([r[0][:, j] for r in res], [r[1][:, j] for r in res], k)
        assert np.array_equal(mp, rp)
        assert_same_f32(ms, rs, "merged scores")


# ---------------------------------------------------------------------------------------------------
# BASELINE.json config 2 size (100k passages): size-independent properties + a sampled oracle check
# ---------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def big():
    idx = synthetic.make_index(seed=31, n_docs=100_000)           # K = 32768 by the reference's rule
    assert idx["centroids"].shape[1] == 32768
    Qs = synthetic.make_queries(idx, 32, 8)
    s = clb.Searcher(index=idx)
    yield idx, Qs, s
    s.close()


def test_full_size_properties(big, oracle):
    idx, Qs, s = big
    k = 1000
    results = {}
    for mode in (0, 1):
        try:
            s.set_mode(mode)
        except clb.Unsupported:
            continue
        pids, scores, n = s.search_batch(Qs, k)
        assert np.all(np.diff(scores, axis=0) <= 0)                               # sorted by score
        ties = np.diff(scores, axis=0) == 0
        assert np.all(np.diff(pids, axis=0)[ties] > 0)                            # ties by ascending pid
        for j in range(Qs.shape[2]):
            assert np.unique(pids[:, j]).size == k
            assert set(pids[:, j]) <= set(s.retrieve(Qs[:, :, j]))                # top-k subset of candidates
        p2, s2, _ = s.search_batch(Qs, k)                                         # idempotent
        assert np.array_equal(pids, p2) and np.array_equal(bits(scores), bits(s2))
        pk, sk, _ = s.search_batch(Qs, 10)                                        # top-10 is a prefix of top-1000
        assert np.array_equal(pk, pids[:10]) and np.array_equal(bits(sk), bits(scores[:10]))
        results[mode] = (pids, scores)
    if len(results) == 2:                                                         # both modes: identical output
        assert np.array_equal(results[0][0], results[1][0])
        assert np.array_equal(bits(results[0][1]), bits(results[1][1]))
    # the returned scores equal the oracle's maxsim of those passages (checks scoring at full size)
    pids, scores = next(iter(results.values()))
    emb_off = np.concatenate([[0], np.cumsum(idx["doclens"])])
    for j in (0, 5):
        sel = pids[:50, j]
        cols = np.concatenate([np.arange(emb_off[p - 1], emb_off[p]) for p in sel])
        D = oracle.decompress(128, 2, idx["centroids"], idx["bucket_weights"], idx["codes"][cols], idx["residuals"][:, cols])
        ref = oracle.maxsim(Qs[:, :, j], D, np.arange(1, 51), idx["doclens"][sel - 1])
        assert_same_f32(scores[:50, j], ref, "full-size score check")


def test_full_size_against_oracle(big, oracle):
    idx, Qs, s = big
    for mode in (0, 1):
        try:
            s.set_mode(mode)
        except clb.Unsupported:
            continue
        for j in (1, 6):
            rp, rs, rn = oracle.search(idx, Qs[:, :, j], 2, 1000)
            pids, scores = s.search_embeddings(Qs[:, :, j], 1000)
            assert s.last_num_candidates == rn
            assert np.array_equal(pids, rp)
            assert_same_f32(scores, rs, f"100k scores mode={mode}")


# ---------------------------------------------------------------------------------------------------
# index() end to end on the device, then Searcher(index_path) + search (BASELINE config 2, scaled down)
# ---------------------------------------------------------------------------------------------------
def test_index_then_search_end_to_end(oracle, tmp_path):
    embs, doclens = synthetic.make_embeddings(seed=51, n_docs=3000, n_components=256)
    enc = clb.PrecomputedEncoder(embs, doclens)
    cfg = clb.ColBERTConfig(index_path=str(tmp_path / "idx"), chunksize=1000, kmeans_niters=4, nbits=2)
    idxr = clb.Indexer(cfg, encoder=enc, collection=list(range(3000)), seed=3)
    assert clb.index(idxr) == cfg.index_path
    assert clb.index(idxr) is None                                  # indexing.jl:64-67: directory exists -> skip
    from colbert_jl_amd import storage
    assert storage.check_all_files_are_saved(cfg.index_path)
    plan = storage.load_json(cfg.index_path, "plan.json")
    assert plan["num_chunks"] == 3 and plan["num_embeddings"] == int(doclens.sum())
    assert plan["embeddings_offsets"][0] == 1
    idx = storage.load_index(cfg.index_path)
    K = plan["num_partitions"]
    assert idx["centroids"].shape == (128, K) and np.array_equal(idx["doclens"], doclens)
    # every stage's output equals the oracle's on the same inputs (stage-wise parity)
    rc, rr = oracle.compress(idx["centroids"], idx["bucket_cutoffs"], 128, 2, embs)
    assert np.array_equal(idx["codes"], rc) and np.array_equal(idx["residuals"], rr)
    rivf, rlens = oracle.build_ivf(idx["codes"], K)
    assert np.array_equal(idx["ivf"], rivf) and np.array_equal(idx["ivf_lengths"], rlens)
    sample = storage._load(cfg.index_path + "/sample"); held = storage._load(cfg.index_path + "/sample_heldout")
    rcut, rw, ravg, _ = oracle.compute_avg_residuals(2, idx["centroids"], held)
    assert_same_f32(idx["bucket_cutoffs"], rcut, "cutoffs"); assert_same_f32(idx["bucket_weights"], rw, "weights")
    assert np.isclose(float(idx["avg_residual"]), ravg, rtol=1e-5)
    assert sample.shape[0] == 128 and held.shape[1] == codec.heldout_size(sample.shape[1] + held.shape[1])
    # search through Searcher(index_path): queries near passages of the collection
    s = clb.Searcher(cfg.index_path)
    off = np.concatenate([[0], np.cumsum(doclens)])
    rng = np.random.default_rng(52)
    for p in (5, 1234, 2999):
        cols = off[p] + rng.integers(0, doclens[p], size=32)
        Q = np.asfortranarray(embs[:, cols])
        pids, scores = clb.search(s, Q, 20)
        rp, rs, _ = oracle.search(idx, Q, nprobe=cfg.nprobe, k=20)
        assert np.array_equal(pids, rp) and pids[0] == p + 1       # the passage itself wins
        assert_same_f32(scores, rs, "end-to-end scores")
    s.close()


# ---------------------------------------------------------------------------------------------------
# multi-GPU index build: the shard-level entry points vs the oracle's sharded restatement (one GPU, two shards)
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dim,n0,n1,K,bsize", [(128, 2600, 1900, 48, 1000), (128, 700, 1, 64, 100), (16, 300, 450, 9, 1000)])
def test_sharded_kmeans_matches_oracle(oracle, dim, n0, n1, K, bsize):
    rng = np.random.default_rng(dim + n0)
    X = np.asfortranarray(rng.standard_normal((dim, n0 + n1)).astype(np.float32))
    X /= np.linalg.norm(X, axis=0, keepdims=True)
    shards = [np.asfortranarray(X[:, :n0]), np.asfortranarray(X[:, n0:])]
    c = np.asfortranarray(X[:, rng.choice(n0 + n1, K, replace=False)])
    hs = [codec.KMeansShard(x, K, bsize) for x in shards]
    cr = c.copy(order="F")
    for _ in range(4):
        parts = [h.pass_(c, want_assignments=True) for h in hs]
        refs = [oracle.kmeans_shard_pass(x, cr, bsize) for x in shards]
        for (s, n, a), (rs, rn, ra) in zip(parts, refs):
            assert np.array_equal(a, ra) and np.array_equal(n, rn)
            assert_same_f32(s, rs, "shard sums")
        gs = np.stack([p[0].ravel(order="F") for p in parts]); gc = np.stack([p[1] for p in parts])
        c, d, conv = codec.kmeans_reduce_update(c, gs, gc)
        cr, rd, rconv = oracle.kmeans_reduce_update(cr, [p[0] for p in refs], [p[1] for p in refs])
        assert conv == rconv and bits(np.float32(d)) == bits(np.float32(rd))
        assert_same_f32(c, cr, "sharded centroids")
    # one shard == clb_kmeans
    one = codec.KMeansShard(X, K, bsize)
    c1 = np.asfortranarray(X[:, :K].copy())
    for _ in range(2):
        s, n = one.pass_(c1)
        c1, _, _ = codec.kmeans_reduce_update(c1, s.ravel(order="F")[None, :], n[None, :])
    ref, _, _ = codec.kmeans(X, np.asfortranarray(X[:, :K].copy()), max_iters=2, point_bsize=bsize)
    assert_same_f32(c1, ref, "one shard vs clb_kmeans")
    for h in hs + [one]:
        h.close()


# ---------------------------------------------------------------------------------------------------
# general shapes: everything the reference accepts (residual.jl:698-721, searching.jl:93-128, utils.jl:327-332)
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dim,nbits", [(64, 2), (128, 8), (24, 8), (256, 4), (8, 1)])
def test_search_general_dim_and_nbits(oracle, dim, nbits):
    """dim != 128 and nbits = 8 take the general-shape path (generic_kernels.hpp): same pids, bit-identical scores."""
    idx = synthetic.make_index(seed=41 + dim + nbits, n_docs=900, K=96, dim=dim, nbits=nbits, doclen_mean=30, doclen_std=12)
    Qs = synthetic.make_queries(idx, 42, 3, T=20)
    check_search(oracle, idx, Qs, k=40, modes=(0,))
    check_search(oracle, idx, Qs, k=7, nprobe=5, modes=(0,))


def test_search_long_queries_large_nprobe_large_k(oracle):
    """On the tuned shape (dim 128, nbits 2): T > 128 goes to the general path; nprobe > 32 selects by a stable sort;
    k up to 16 384 is sorted by the one-work-group top-k kernel (128 KB of LDS), beyond that by a full stable sort -- in
    both modes -- and nprobe = K makes every passage a candidate."""
    idx = synthetic.make_index(seed=43, n_docs=6000, K=256, doclen_mean=24, doclen_std=6)
    check_search(oracle, idx, synthetic.make_queries(idx, 44, 2, T=150), k=30, modes=(0,))
    Qs = synthetic.make_queries(idx, 45, 3)
    check_search(oracle, idx, Qs, k=50, nprobe=40)
    check_search(oracle, idx, Qs, k=5000, nprobe=64)
    s = clb.Searcher(index=idx)
    try:
        pids, scores = s.search_embeddings(Qs[:, :, 0], 6000, nprobe=256)           # every passage, fully ranked
        rp, rs, rn = oracle.search(idx, Qs[:, :, 0], nprobe=256, k=6000)
        assert rn == 6000 and np.array_equal(pids, rp)
        assert_same_f32(scores, rs, "nprobe = K, k = n_docs")
        with pytest.raises(clb.BoundsError):
            s.search_embeddings(Qs[:, :, 0], 6001, nprobe=256)
    finally:
        s.close()
    big = synthetic.make_index(seed=46, n_docs=20000, K=256, doclen_mean=10, doclen_std=2)
    Qb = synthetic.make_queries(big, 47, 2)
    check_search(oracle, big, Qb, k=12000, nprobe=64)          # the largest power of two the top-k kernel sorts: 16 384
    check_search(oracle, big, Qb, k=17000, nprobe=128)         # past it: the full stable sort


# ---------------------------------------------------------------------------------------------------
# the library's own radix sort and scan (csrc/sort.hip; rocPRIM until round 5) against numpy's stable sort
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n", [1, 63, 64, 8191, 8192, 8193, 100_003, 2_500_001])
def test_radix_sort_is_numpys_stable_sort(n):
    """Stable LSD radix sort: (key, value) pairs on the low end_bit bits of uint32 keys (what sortperm(codes) needs:
    collection_indexer.jl:350 -- equal codes keep their embedding order), uint64 pairs and keys, float keys; sizes around the
    tile edges (8 192 elements per work-group, 2 048 per wave); few distinct keys (long runs of equal digits) and many."""
    import ctypes as C
    l = clb.lib()
    rng = np.random.default_rng(n)
    for end_bit, hi in ((18, 1 << 18), (32, 1 << 32), (7, 100), (32, 3)):
        keys = rng.integers(0, hi, size=n, dtype=np.uint64).astype(np.uint32)
        vals = np.arange(n, dtype=np.uint32)
        ko, vo = np.empty_like(keys), np.empty_like(vals)
        assert l.clb_debug_sort(0, 32, C.c_void_p(keys.ctypes.data), C.c_void_p(vals.ctypes.data), C.c_int64(n), end_bit,
                                C.c_void_p(ko.ctypes.data), C.c_void_p(vo.ctypes.data)) == 0, l.clb_last_error()
        order = np.argsort(keys & np.uint32((1 << end_bit) - 1 if end_bit < 32 else 0xffffffff), kind="stable")
        assert np.array_equal(vo, order.astype(np.uint32)) and np.array_equal(ko, keys[order]), (n, end_bit, hi)
    k64 = rng.integers(0, 1 << 63, size=n, dtype=np.uint64) if n % 2 else (rng.integers(0, 50, size=n, dtype=np.uint64) << np.uint64(40))
    vals = rng.integers(0, 1 << 32, size=n, dtype=np.uint64).astype(np.uint32)
    ko, vo = np.empty_like(k64), np.empty_like(vals)
    assert l.clb_debug_sort(0, 64, C.c_void_p(k64.ctypes.data), C.c_void_p(vals.ctypes.data), C.c_int64(n), 64,
                            C.c_void_p(ko.ctypes.data), C.c_void_p(vo.ctypes.data)) == 0, l.clb_last_error()
    order = np.argsort(k64, kind="stable")
    assert np.array_equal(ko, k64[order]) and np.array_equal(vo, vals[order])
    assert l.clb_debug_sort(0, 64, C.c_void_p(k64.ctypes.data), None, C.c_int64(n), 64, C.c_void_p(ko.ctypes.data), None) == 0
    assert np.array_equal(ko, np.sort(k64))
    f = rng.normal(size=n).astype(np.float32)
    f[rng.integers(0, n, size=max(1, n // 50))] = np.float32(0.0)
    f[rng.integers(0, n, size=max(1, n // 70))] = np.float32(-0.0)
    if n > 10:
        f[:3] = [np.float32(np.inf), np.float32(-np.inf), np.float32(1e-42)]          # infinities and a subnormal
    fo = np.empty_like(f)
    assert l.clb_debug_sort(0, -32, C.c_void_p(f.ctypes.data), None, C.c_int64(n), 32, C.c_void_p(fo.ctypes.data), None) == 0
    assert np.array_equal(fo, np.sort(f))                                              # values (-0.0 == 0.0 compare equal)
    assert np.all((fo[:-1] != fo[1:]) | (np.signbit(fo[:-1]) >= np.signbit(fo[1:])))  # ... and -0.0 in front of +0.0


@pytest.mark.parametrize("n", [0, 1, 1000, 16383, 16384, 16385, 300_000, 5_000_000])
def test_exclusive_scan_matches_numpy(n):
    import ctypes as C
    l = clb.lib()
    rng = np.random.default_rng(n + 1)
    x = rng.integers(0, 1 << 12, size=n, dtype=np.uint64).astype(np.uint32)
    out = np.empty(n + 1, dtype=np.uint32)
    assert l.clb_debug_exclusive_scan(0, C.c_void_p(x.ctypes.data) if n else None, C.c_int64(n), C.c_void_p(out.ctypes.data)) == 0, l.clb_last_error()
    want = (np.concatenate([np.zeros(1, np.uint64), np.cumsum(x.astype(np.uint64))]) & np.uint64(0xffffffff)).astype(np.uint32)
    assert np.array_equal(out, want)
