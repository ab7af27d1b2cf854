"""Pins the CPU oracle against every known-answer vector the reference's own tests hold for the hot
path (tests/golden/reference_kats.json, transcribed by tests/golden/make_reference_kats.py), and
re-states the reference's property tests (round trips, shapes, error contracts) with our own RNG.
Each test names the reference test it mirrors."""
import numpy as np
import pytest


def F(shape, flat, dtype):
    return np.array(flat, dtype=dtype).reshape(shape, order="F")


# ---- test/indexing/codecs/residual.jl -------------------------------------------------------------
def test_binarize_kats(oracle, kats):
    for c in kats["_binarize"]["cases"]:
        got = oracle.binarize_bits(np.array(c["data"]), c["nbits"])
        assert got.shape == tuple(c["shape"])
        assert np.array_equal(got, F(c["shape"], c["expected_flat"], bool))
    de = kats["_binarize"]["domain_error"]
    with pytest.raises(oracle.DomainError):
        oracle.binarize_bits(np.array(de["data"]), de["nbits"])


def test_unbinarize_kats(oracle, kats):
    for c in kats["_unbinarize"]["cases"]:
        got = oracle.unbinarize(F(c["shape"], c["bits_flat"], bool))
        assert np.array_equal(got, np.array(c["expected"]))
    rng = np.random.default_rng(0)
    nbits = 7
    assert np.all(oracle.unbinarize(np.zeros((nbits, 5, 3), bool)) == 0)
    assert np.all(oracle.unbinarize(np.ones((nbits, 5, 3), bool)) == (1 << nbits) - 1)
    data = rng.integers(0, 1 << 11, size=(13, 9))           # "_unbinarize inverts _binarize" :154-161
    assert np.array_equal(oracle.unbinarize(oracle.binarize_bits(data, 11)), data)


def test_bucket_indices_kats(oracle, kats):
    for c in kats["_bucket_indices"]["cases"]:
        got = oracle.bucket_indices(np.array(c["data"], dtype=np.float32), c["cutoffs"])
        assert np.array_equal(got, np.array(c["expected"]))
    rng = np.random.default_rng(1)
    cut = np.sort(rng.random(17).astype(np.float32))
    got = oracle.bucket_indices(rng.random((6, 7)).astype(np.float32), cut)
    assert got.min() >= 0 and got.max() <= cut.size


def test_packbits_kats(oracle, kats):
    k = kats["_packbits"]
    for c in k["cases"]:
        got = oracle.packbits(F(c["shape"], c["bits_flat"], bool))
        assert np.array_equal(got.ravel(order="F"), np.array(c["expected_bytes"], dtype=np.uint8))
    assert np.all(oracle.packbits(np.zeros((3, 16, 4), bool)) == 0)
    assert np.all(oracle.packbits(np.ones((3, 16, 4), bool)) == 0xFF)
    alt = np.ones(3 * 16 * 4, bool); alt[1::2] = False          # :248-253
    assert np.all(oracle.packbits(alt.reshape((3, 16, 4), order="F")) == k["alternating_byte"])
    with pytest.raises(oracle.DomainError):
        oracle.packbits(np.ones(tuple(k["domain_error_shape"]), bool))
    out = oracle.packbits(np.ones((5, 24, 7), bool))
    assert out.dtype == np.uint8 and out.shape == (24 * 5 // 8, 7)


def test_unpackbits_kat(oracle, kats):
    k = kats["_unpackbits"]
    packed = F(k["packed_shape"], k["packed_flat"], np.uint8)
    got = oracle.unpackbits(packed, k["nbits"])
    assert got.shape == tuple(k["expected_shape"])
    assert np.array_equal(got, F(k["expected_shape"], k["expected_flat"], bool))
    assert not oracle.unpackbits(np.zeros((6, 5), np.uint8), 3).any()
    assert oracle.unpackbits(np.full((6, 5), 0xFF, np.uint8), 3).all()
    with pytest.raises(oracle.DomainError):
        oracle.unpackbits(np.zeros((7, 5), np.uint8), 3)
    rng = np.random.default_rng(2)                                # "_unpackbits inverts _packbits"
    for nbits in (1, 2, 3, 4, 8):
        bits = rng.random((nbits, 8 * 5, 6)) < 0.5
        assert np.array_equal(oracle.unpackbits(oracle.packbits(bits), nbits), bits)


def test_binarize_contract(oracle):
    rng = np.random.default_rng(3)
    dim, nbits = 24, 3
    cut = np.sort(rng.random((1 << nbits) - 1).astype(np.float32))
    res = rng.random((dim, 11)).astype(np.float32)
    out = oracle.binarize(dim, nbits, cut, res)
    assert out.dtype == np.uint8 and out.shape == (dim // 8 * nbits, 11)
    with pytest.raises(oracle.DomainError):
        oracle.binarize(7, 7, np.sort(rng.random(127).astype(np.float32)), rng.random((7, 10)).astype(np.float32))
    with pytest.raises(oracle.DomainError):
        oracle.binarize(8, 8, np.sort(rng.random(254).astype(np.float32)), rng.random((8, 10)).astype(np.float32))


def test_decompress_residuals_inverts_binarize(oracle):
    rng = np.random.default_rng(4)                                # :975-991
    for nbits in (1, 2, 4):
        dim = 40
        cut = np.sort(rng.random((1 << nbits) - 1).astype(np.float32))
        w = np.sort(rng.random(1 << nbits).astype(np.float32))
        res = rng.random((dim, 37)).astype(np.float32)
        idx = oracle.bucket_indices(res, cut)
        got = oracle.decompress_residuals(dim, nbits, w, oracle.binarize(dim, nbits, cut, res))
        assert np.array_equal(got, w[idx])
    with pytest.raises(oracle.DomainError):
        oracle.decompress_residuals(8, 8, np.zeros(255, np.float32), np.zeros((8, 3), np.uint8))
    with pytest.raises(oracle.DomainError):
        oracle.decompress_residuals(8, 8, np.zeros(256, np.float32), np.zeros((7, 3), np.uint8))


def test_nbits2_byte_layout(oracle):
    """for nbits=2, byte[j] = sum_i idx[4j+i] << 2i."""
    rng = np.random.default_rng(5)
    dim = 128
    idx = rng.integers(0, 4, size=(dim, 3))
    packed = oracle.packbits(oracle.binarize_bits(idx, 2))
    ref = (idx.reshape(32, 4, 3) << (2 * np.arange(4))[None, :, None]).sum(axis=1).astype(np.uint8)
    assert np.array_equal(packed, ref)


def test_compress_into_codes(oracle):
    rng = np.random.default_rng(6)                                # :5-57
    embs = rng.random((17, 1)).astype(np.float32)
    assert np.array_equal(oracle.compress_into_codes(embs, embs), [1])
    embs = oracle.normalize_array(rng.random((33, 14)).astype(np.float32))
    perm = rng.permutation(14)
    codes = oracle.compress_into_codes(embs[:, perm], embs)
    assert np.array_equal(codes, np.argsort(perm) + 1)            # sortperm(perm)
    sub = rng.permutation(14)[:6]
    codes = oracle.compress_into_codes(embs[:, sub], embs)
    assert codes.min() >= 1 and codes.max() <= 6
    assert np.array_equal(codes[sub], np.arange(1, 7))
    cent = oracle.normalize_array(rng.random((20, 9)).astype(np.float32))
    mapping = rng.integers(0, 9, size=27)
    noisy = cent[:, mapping] + (rng.random(27).astype(np.float32) * 2e-5 - 1e-5)[None, :]
    assert np.array_equal(oracle.compress_into_codes(cent, noisy), mapping + 1)
    with pytest.raises(oracle.DimensionMismatch):
        oracle.compress_into_codes(cent, noisy, n_codes=5)


def test_compress_structure(oracle):
    rng = np.random.default_rng(7)                                # :870-949
    dim, nbits = 32, 2
    cut = np.sort(rng.random(3).astype(np.float32))               # all > 0: zero residual -> bucket 0
    embs = oracle.normalize_array(rng.random((dim, 12)).astype(np.float32))
    perm = rng.permutation(12)
    codes, res = oracle.compress(embs[:, perm], cut, dim, nbits, embs)
    assert np.array_equal(codes, np.argsort(perm) + 1) and not res.any()
    assert codes.dtype == np.uint32 and res.dtype == np.uint8 and res.shape == (dim // 8 * nbits, 12)
    sub = rng.permutation(12)[:5]
    codes, res = oracle.compress(embs[:, sub], cut, dim, nbits, embs)
    assert np.array_equal(codes[sub], np.arange(1, 6)) and not res[:, sub].any()


def test_decompress_shape_and_values(oracle):
    rng = np.random.default_rng(8)                                # :993-1007 (+ values, unpinned upstream)
    dim, nbits, n, K = 128, 2, 50, 19
    w = np.sort(rng.normal(0, 0.03, 4).astype(np.float32))
    cent = rng.normal(size=(dim, K)).astype(np.float32)
    codes = rng.integers(1, K + 1, size=n).astype(np.uint32)
    res = rng.integers(0, 256, size=(dim // 4, n)).astype(np.uint8)
    D = oracle.decompress(dim, nbits, cent, w, codes, res)
    assert D.dtype == np.float32 and D.shape == (dim, n)
    # independent numpy restatement of residual.jl:759-784 (float64 accumulation -> tolerance)
    idx = np.stack([(res >> (2 * i)) & 3 for i in range(4)], axis=1).reshape(dim, n)
    x = cent[:, codes - 1] + w[idx]
    ref = x / (np.sqrt((x.astype(np.float64) ** 2).sum(axis=0)) + np.finfo(np.float32).eps)
    assert np.allclose(D, ref, atol=1e-6)
    with pytest.raises(oracle.DomainError):
        oracle.decompress(dim, nbits, cent, w, codes[:-1], res)
    bad = codes.copy(); bad[0] = K + 1
    with pytest.raises(oracle.DomainError):
        oracle.decompress(dim, nbits, cent, w, bad, res)


# ---- test/search/ranking.jl, test/searching.jl ---------------------------------------------------------
def test_cids_to_eids(oracle, kats):
    k = kats["_cids_to_eids!"]
    for c in k["cases"]:
        got = oracle.cids_to_eids(c["n_eids"], c["centroid_ids"], c["ivf"], c["ivf_lengths"])
        assert np.array_equal(got, c["expected"])
    for c in k["dimension_mismatch"]:
        with pytest.raises(oracle.DimensionMismatch):
            oracle.cids_to_eids(c["n_eids"], c["centroid_ids"], c["ivf"], c["ivf_lengths"])
    assert oracle.cids_to_eids(0, [], [], []).size == 0
    assert oracle.cids_to_eids(0, [], [1, 2, 3, 4, 5, 6], [3, 2, 1]).size == 0
    rng = np.random.default_rng(9)                                # Test 2: random partitioning
    n, K = 500, 13
    assign = rng.integers(0, K, size=n)
    lists = [rng.permutation(np.nonzero(assign == c)[0] + 1) for c in range(K)]
    ivf = np.concatenate(lists); lens = np.array([len(x) for x in lists])
    cids = rng.permutation(K)[:7] + 1
    got = oracle.cids_to_eids(int(lens[cids - 1].sum()), cids, ivf, lens)
    assert np.array_equal(got, np.concatenate([lists[c - 1] for c in cids]))


def test_retrieve_kat(oracle, kats):
    k = kats["retrieve"]
    got = oracle.retrieve(k["ivf"], k["ivf_lengths"], np.array(k["centroids"], np.float32), k["emb2pid"],
                          k["nprobe"], np.array(k["Q"], np.float32))
    assert np.array_equal(got, k["expected_pids"])


def test_collect_compressed(oracle, kats):
    for c in kats["_collect_compressed_embs_for_pids"]["cases"]:
        res = np.array(c["residuals"], dtype=np.uint8).reshape(2, -1)
        oc, orr = oracle.collect_compressed_embs_for_pids(c["doclens"], c["codes"], res, c["pids"])
        assert np.array_equal(oc, np.array(c["expected_codes"], dtype=np.uint32))
        assert np.array_equal(orr, np.array(c["expected_residuals"], dtype=np.uint8).reshape(2, -1))
        assert oc.dtype == np.uint32 and orr.dtype == np.uint8


def test_maxsim_kat(oracle, kats):
    k = kats["maxsim"]
    got = oracle.maxsim(np.array(k["Q"], np.float32), np.array(k["D"], np.float32), k["pids"], k["doclens"])
    assert np.array_equal(got, np.array(k["expected_scores"], np.float32))
    dm = k["dimension_mismatch"]
    with pytest.raises(oracle.DimensionMismatch):
        oracle.maxsim(np.array(dm["Q"], np.float32), np.array(dm["D"], np.float32), dm["pids"], dm["doclens"])
    rng = np.random.default_rng(10)
    doclens = rng.integers(1, 11, size=200)
    Q = rng.random((128, 100)).astype(np.float32); D = rng.random((128, int(doclens.sum()))).astype(np.float32)
    s = oracle.maxsim(Q, D, np.arange(1, 201), doclens)
    assert s.shape == (200,) and s.dtype == np.float32
    off = np.concatenate([[0], np.cumsum(doclens)])
    S = Q.astype(np.float64).T @ D.astype(np.float64)
    ref = np.array([S[:, off[i]:off[i + 1]].max(axis=1).sum() for i in range(200)])
    assert np.allclose(s, ref, rtol=1e-5)


def test_build_emb2pid(oracle, kats):
    for c in kats["_build_emb2pid"]["cases"]:
        assert np.array_equal(oracle.build_emb2pid(c["doclens"]), c["expected"])
    assert np.array_equal(oracle.build_emb2pid([7] * 5), np.repeat(np.arange(1, 6), 7))


# ---- test/utils.jl -----------------------------------------------------------------------------------
def test_topk_kat(oracle, kats):
    k = kats["_topk"]
    data = np.array(k["data"], np.float32)
    assert np.array_equal(oracle.topk(data, k["k"], dims=1), k["expected_dims1"])
    assert np.array_equal(oracle.topk(data, k["k"], dims=2), k["expected_dims2"])
    with pytest.raises(oracle.DomainError):
        oracle.topk(data, k["k"], dims=3)


def test_kmeans_pieces(oracle, kats):
    k = kats["compute_distances_kernel!"]["single"]
    got = oracle.compute_distances_kernel(np.zeros((1, 1)), np.array(k["batch_data"]), np.array(k["centroids"]))
    assert np.allclose(got, k["expected"])
    assert not oracle.compute_distances_kernel(np.zeros((4, 6)), np.ones((9, 6)), np.ones((9, 4))).any()
    dim, b = 11, 9                                                # Test 3: dist[j,i] == dim*(i-j)^2 exactly
    data = np.ones((dim, b), np.float32) * np.arange(1, b + 1, dtype=np.float32)
    got = oracle.compute_distances_kernel(np.zeros((b, b)), data, data)
    ij = np.arange(1, b + 1)
    assert np.array_equal(got, (dim * (ij[:, None] - ij[None, :]) ** 2).astype(np.float32))
    with pytest.raises(oracle.DimensionMismatch):
        oracle.compute_distances_kernel(np.zeros((3, 2)), np.ones((2, 2)), np.ones((2, 2)))
    with pytest.raises(oracle.DimensionMismatch):
        oracle.compute_distances_kernel(np.zeros((3, 2)), np.ones((2, 2)), np.ones((3, 3)))
    rng = np.random.default_rng(11)                               # assign_clusters_kernel! :90-105
    dist = np.stack([rng.permutation(40).astype(np.float32) + 1 for _ in range(30)], axis=1)
    assert np.array_equal(oracle.assign_clusters_kernel(30, dist), dist.argmin(axis=0) + 1)
    with pytest.raises(oracle.DimensionMismatch):
        oracle.assign_clusters_kernel(1, np.array([[1.0, 2.0], [4.0, 5.0]]))
    oh = kats["onehot_encode!"]                                   # :117-129
    assert np.array_equal(oracle.onehot_encode(np.zeros((4, 4)), oh["assignments"], oh["k"]), oh["expected"])
    assert np.array_equal(oracle.onehot_encode(np.zeros((1, 3)), [1, 1, 1], 1), [[1, 1, 1]])
    with pytest.raises(oracle.DimensionMismatch):
        oracle.onehot_encode(np.zeros((3, 3)), [1, 2], 3)
    K, n = 6, 14                                                  # update_centroids_kernel! :53-88
    p2c = rng.integers(0, K, size=n)
    onehot = np.zeros((K, n), np.float32); onehot[p2c, np.arange(n)] = 1
    got = oracle.update_centroids_kernel(np.ones((5, K)), np.ones((5, n)), onehot)
    assert np.array_equal(got, 1 + np.bincount(p2c, minlength=K)[None, :] * np.ones((5, 1)))
    with pytest.raises(oracle.DimensionMismatch):
        oracle.update_centroids_kernel(np.zeros((3, 2)), np.ones((2, 2)), np.eye(2))
    with pytest.raises(oracle.DimensionMismatch):
        oracle.update_centroids_kernel(np.zeros((2, 2)), np.ones((2, 2)), np.ones((2, 3)))


def test_kmeans_fixed_point(oracle):
    rng = np.random.default_rng(12)                               # test/utils.jl:138-145
    data = rng.random((24, 31)).astype(np.float32)
    perm = rng.permutation(31)
    cent, ids, _ = oracle.kmeans(data, data[:, perm], max_iters=10)
    assert np.array_equal(cent[:, ids - 1], data)


def test_normalize_array(oracle):
    rng = np.random.default_rng(13)
    X = oracle.normalize_array(rng.random((37, 21)).astype(np.float32), dims=1)
    assert np.allclose(np.linalg.norm(X, axis=0), 1, atol=1e-5)
    X = oracle.normalize_array(rng.random((37, 21)).astype(np.float32), dims=2)
    assert np.allclose(np.linalg.norm(X, axis=1), 1, atol=1e-5)
    Z = oracle.normalize_array(np.zeros((8, 2), np.float32))      # zero columns stay zero: 0/(0+eps)
    assert not Z.any()


# ---- test/indexing/collection_indexer.jl ---------------------------------------------------------------
def test_bucket_cutoffs_and_weights(oracle, kats):
    k = kats["_bucket_cutoffs_and_weights"]
    cut, w = oracle.bucket_cutoffs_and_weights(k["nbits"], np.array(k["heldout_avg_residual"], np.float32))
    assert np.allclose(cut, k["expected_cutoffs"]) and np.allclose(w, k["expected_weights"])
    v = np.float32(0.3712)
    cut, w = oracle.bucket_cutoffs_and_weights(3, np.full((4, 5), v, np.float32))
    assert np.all(cut == v) and np.all(w == v)
    rng = np.random.default_rng(14)
    x = rng.normal(size=1000).astype(np.float32)
    cut, w = oracle.bucket_cutoffs_and_weights(2, x)
    assert np.allclose(cut, np.quantile(x.astype(np.float64), [0.25, 0.5, 0.75]), atol=1e-6)
    assert np.allclose(w, np.quantile(x.astype(np.float64), [0.125, 0.375, 0.625, 0.875]), atol=1e-6)


def test_compute_avg_residuals(oracle):
    rng = np.random.default_rng(15)
    cent = oracle.normalize_array(rng.normal(size=(16, 7)).astype(np.float32))
    held = oracle.normalize_array(rng.normal(size=(16, 60)).astype(np.float32))
    cut, w, avg, codes = oracle.compute_avg_residuals(2, cent, held)
    assert np.array_equal(codes, (held.T @ cent).argmax(axis=1) + 1)
    res = held - cent[:, codes - 1]
    assert np.isclose(avg, np.abs(res).mean(axis=1).mean(), rtol=1e-5)
    assert np.all(np.diff(cut) >= 0) and np.all(np.diff(w) >= 0)
    with pytest.raises(oracle.DimensionMismatch):
        oracle.compute_avg_residuals(2, cent, held, n_codes=3)


def test_collect_embedding_id_offset(oracle, kats):
    for c in kats["_collect_embedding_id_offset"]["cases"]:
        tot, off = oracle.collect_embedding_id_offset(c["counts"])
        assert tot == c["total"] and np.array_equal(off, c["offsets"])


def test_build_ivf_kat(oracle, kats):
    k = kats["_build_ivf"]
    ivf, lens = oracle.build_ivf(k["codes"], k["num_partitions"])
    assert np.array_equal(ivf, k["expected_ivf"]) and np.array_equal(lens, k["expected_ivf_lengths"])
    rng = np.random.default_rng(16)
    codes = rng.integers(1, 301, size=5000).astype(np.uint32)
    ivf, lens = oracle.build_ivf(codes, 300)
    assert np.array_equal(ivf, np.argsort(codes, kind="stable") + 1)
    assert np.array_equal(lens, np.bincount(codes, minlength=301)[1:])


def test_setup_sizing(oracle, kats):
    for c in kats["setup_sizing"]["cases"]:
        plan = oracle.setup(c["num_documents"], c["avg_doclen_est"], 10 ** 9, 25000, 1)
        assert plan["num_partitions"] == c["num_partitions"]
        if "num_embeddings_est" in c:
            assert np.float32(plan["num_embeddings_est"]) == np.float32(c["num_embeddings_est"])
    plan = oracle.setup(100, 50.0, 37, None, 2)                   # test/indexing/collection_indexer.jl:38-83
    assert plan["num_partitions"] == 37 and plan["chunksize"] == 51 and plan["num_chunks"] == 2
    assert oracle.num_sampled_pids(10) == 10 and oracle.num_sampled_pids(141431) == 65915
    assert oracle.heldout_size(10 ** 7) == 50000 and oracle.heldout_size(3) == 1


# ---- encoder epilogue: test/modelling/embedding_utils.jl -------------------------------------------------
def test_doc_epilogue(oracle):
    rng = np.random.default_rng(17)
    dim, L, N = 16, 9, 4
    D = rng.normal(size=(dim, L, N)).astype(np.float32)
    ids = rng.integers(1, 30, size=(L, N)).astype(np.int32)
    skip = [3, 7, 11]
    out, doclens = oracle.doc_epilogue(D, ids, skip)
    mask = ~np.isin(ids, skip)
    assert np.array_equal(doclens, mask.sum(axis=0))
    flat = D.reshape(dim, L * N, order="F")[:, mask.ravel(order="F")]
    ref = flat / (np.linalg.norm(flat.astype(np.float64), axis=0) + np.finfo(np.float32).eps)
    assert out.shape == ref.shape and np.allclose(out, ref, atol=1e-6)
    Q = oracle.query_epilogue(D, ids, [999])
    assert np.allclose(np.linalg.norm(Q.reshape(dim, -1, order="F"), axis=0), 1, atol=1e-5)


# ---- whole search: searching.jl:102-127 (unpinned upstream; cross-checked against numpy) -------------------
def test_search_matches_numpy(oracle):
    from tests.util_synth import tiny_index
    idx, Q = tiny_index(seed=18, n_docs=300, K=64)
    pids, scores, ncand = oracle.search(idx, Q, nprobe=2, k=10)
    # numpy float64 restatement of the same pipeline
    C = idx["centroids"].astype(np.float64); Qd = Q.astype(np.float64)
    cells = Qd.T @ C
    top = np.argsort(-cells, axis=1, kind="stable")[:, :2]
    cids = np.unique(top)
    off = np.concatenate([[0], np.cumsum(idx["ivf_lengths"])])
    eids = np.concatenate([idx["ivf"][off[c]:off[c + 1]] for c in cids])
    emb2pid = oracle.build_emb2pid(idx["doclens"])
    cand = np.unique(emb2pid[eids - 1])
    assert ncand == cand.size
    D = oracle.decompress(128, 2, idx["centroids"], idx["bucket_weights"], idx["codes"], idx["residuals"]).astype(np.float64)
    doff = np.concatenate([[0], np.cumsum(idx["doclens"])])
    sc = np.array([(Qd.T @ D[:, doff[p - 1]:doff[p]]).max(axis=1).sum() for p in cand])
    order = np.argsort(-sc, kind="stable")[:10]
    assert np.array_equal(pids, cand[order])
    assert np.allclose(scores, sc[order], atol=1e-4)
    with pytest.raises(oracle.BoundsError):
        oracle.search(idx, Q, nprobe=2, k=ncand + 1)
