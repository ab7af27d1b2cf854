"""GPU parity at BASELINE.json's full sizes (configs 2, 4, 5) and on adversarial inputs of the two-pass
mode.  Everything goes through the C ABI (ctypes); the CPU oracle is the checker.

  config 2: 100 k passages -- the index-build stages at full size, checked against oracle.compress on a sample
            of the embeddings and against oracle.build_ivf on all codes;
  config 4: 1 M passages, K = 131 072 -- both modes vs the oracle, and the 8-shard two-phase path on one GPU
            (shards one after the other) merged == oracle;
  config 5: 10 M passages on one GPU -- size-independent properties (needs ~100 GB of host memory: skipped when
            the box has less).
"""
import numpy as np
import pytest

import colbert_jl_amd as clb
from colbert_jl_amd import codec, synthetic

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def same_f32(a, b, what=""):
    a = np.asarray(a, dtype=np.float32); b = np.asarray(b, dtype=np.float32)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    assert np.array_equal(bits(a), bits(b)), (what, float(np.max(np.abs(a.astype(np.float64) - b))))


# ---------------------------------------------------------------------------------------------------
# adversarial inputs of the two-pass mode (eps tested on one friendly corpus only)
# ---------------------------------------------------------------------------------------------------
def _adversarial_index(seed, cent_scale, weights):
    idx = synthetic.make_index(seed=seed, n_docs=4000, K=512, constant_doclen=True, doclen_mean=24)
    rng = np.random.default_rng(seed + 1)
    C = idx["centroids"].copy(order="F")
    C *= rng.choice(np.asarray(cent_scale, dtype=np.float32), size=(1, C.shape[1]))     # centroid norms 0.006 .. 50
    idx["centroids"] = np.asfortranarray(C.astype(np.float32))
    idx["bucket_weights"] = np.asarray(weights, dtype=np.float32)
    L = 24
    for p in range(0, 4000, 4):                                   # duplicated passages: ties straddle rank k
        idx["codes"][(p + 1) * L:(p + 2) * L] = idx["codes"][p * L:(p + 1) * L]
        idx["residuals"][:, (p + 1) * L:(p + 2) * L] = idx["residuals"][:, p * L:(p + 1) * L]
    idx["ivf"], idx["ivf_lengths"] = synthetic.build_ivf(idx["codes"], 512)
    return idx


@pytest.mark.parametrize("rows", [0, 1], ids=["fp16_rows", "cell8_rows"])
@pytest.mark.parametrize("case", ["mixed_norms", "big_weights", "unnormalised_q", "huge_q", "tiny_q", "outlier_centroid"])
def test_two_pass_adversarial(oracle, case, rows):
    """Centroid norms from 0.01 to 50, bucket weights +-0.5, un-normalised queries (token norms 1e-3 .. 1e3; and
    1e5, where the fp16 score table would overflow and the guard must route the query to the exact kernel), duplicated
    passages: the observed error stays within the proven bound, and mode 1 == mode 0 == oracle bit for bit.
    rows = 1: the score table of batches of 16+ queries as 32-byte rows of 8-bit cells (clb_searcher_set_score_rows) -- the
    bound check runs the query as sixteen copies of itself, the batch below is the three queries six times over.
    outlier_centroid: ONE centroid of norm 40 among unit ones blows every token's cell range (and step) up 40 times."""
    weights = [-0.041035336, -0.009812315, 0.008938393, 0.039779153]
    scale = [1.0]
    qscale = None
    if case == "mixed_norms":
        scale = [0.01, 0.3, 1.0, 7.0, 50.0]
    elif case == "big_weights":
        weights = [-0.5, -0.11, 0.13, 0.5]
    elif case == "unnormalised_q":
        scale = [0.05, 1.0, 20.0]; qscale = (1e-3, 1e3)
    elif case == "huge_q":
        scale = [0.5, 1.0, 50.0]; qscale = (1e4, 1e5)
    elif case == "tiny_q":
        qscale = (1e-20, 1e-18)
    idx = _adversarial_index(101, scale, weights)
    if case == "outlier_centroid":
        C = idx["centroids"].copy(order="F")
        C[:, 137] *= np.float32(40.0)
        idx["centroids"] = np.asfortranarray(C)
    Qs = synthetic.make_queries(idx, 102, 3)
    if qscale is not None:
        rng = np.random.default_rng(103)
        f = np.exp(rng.uniform(np.log(qscale[0]), np.log(qscale[1]), size=(1, Qs.shape[1], Qs.shape[2])))
        Qs = np.asfortranarray((Qs * f).astype(np.float32))
    k = 60
    s = clb.Searcher(index=idx)
    try:
        s.set_score_rows(rows)
        assert s.score_rows == rows
        for j in range(Qs.shape[2]):
            rp, rs, rn = oracle.search(idx, Qs[:, :, j], nprobe=2, k=k)
            out = {}
            for mode in (0, 1):
                s.set_mode(mode)
                out[mode] = s.search_embeddings(Qs[:, :, j], k)
                assert s.last_num_candidates == rn
            for mode in (0, 1):
                assert np.array_equal(out[mode][0], rp), (case, mode, j)
                same_f32(out[mode][1], rs, f"{case} mode={mode} q={j}")
            d = s.debug_scores(Qs[:, :, j], k)
            if np.isfinite(d["eps"]):                              # guarded queries report eps = inf
                err = np.abs(d["approx"].astype(np.float64) - d["exact"].astype(np.float64))
                assert err.max() <= d["eps"], (case, j, err.max(), d["eps"])
            else:                                                  # everything goes to the exact kernel
                assert (case == "huge_q" or (rows == 1 and case == "tiny_q")) and d["n_rescore"] == rn
        s.set_mode(1)
        Qb = np.asfortranarray(np.concatenate([Qs] * 6, axis=2)) if rows else Qs      # 18 queries: the 16-query centroid kernel
        bp, bs, _ = s.search_batch(Qb, k)
        for j in range(Qb.shape[2]):
            rp, rs, _ = oracle.search(idx, Qs[:, :, j % Qs.shape[2]], nprobe=2, k=k)
            assert np.array_equal(bp[:, j], rp)
            same_f32(bs[:, j], rs, f"{case} batch q={j}")
    finally:
        s.close()


def test_two_phase_heterogeneous_shards(oracle):
    """with a GLOBAL threshold every shard must use the LARGEST eps.  One shard holds passages whose
    embeddings decompress to (almost) the zero vector -- inv_norm = 1/eps32, a huge shard-local eps and wildly wrong
    approximate scores that inflate the global tau; the other shards' own eps is small.  After share_bound_consts the
    merged two-phase result equals mode 0 and the oracle."""
    torch = pytest.importorskip("torch")
    from colbert_jl_amd.distributed import DeviceSearch, merge_packed, share_bound_consts
    from colbert_jl_amd.sharding import shard_index
    idx = synthetic.make_index(seed=111, n_docs=6000, K=256, constant_doclen=True, doclen_mean=20)
    w0 = idx["bucket_weights"][0]
    C = idx["centroids"].copy(order="F")
    C[:, 7] = -w0                                   # centroid 8 + all-zero residual bytes -> c + r == 0 exactly
    idx["centroids"] = C
    L = 20
    for p in range(0, 40):                          # passages 1..40 (all in shard 0): half their embeddings vanish
        idx["codes"][p * L:p * L + 10] = 8
        idx["residuals"][:, p * L:p * L + 10] = 0
    idx["ivf"], idx["ivf_lengths"] = synthetic.build_ivf(idx["codes"], 256)
    Qs = synthetic.make_queries(idx, 112, 4)
    Qs[:, :6, :] = (C[:, [7]] / np.linalg.norm(C[:, 7]))[:, :, None]      # probe centroid 8: those passages are candidates
    Qs = np.asfortranarray(Qs.astype(np.float32))
    Qdev = torch.from_numpy(np.ascontiguousarray(Qs.transpose(2, 1, 0))).cuda()
    world, k, B = 4, 30, Qs.shape[2]
    searchers = []
    for r in range(world):
        sub, off = shard_index(idx, r, world)
        searchers.append(clb.Searcher(index=sub, pid_offset=off))
    consts = [s.bound_consts for s in searchers]
    assert consts[0][2] > 1e5 * consts[1][2]                      # shard 0 alone sees the degenerate embeddings
    share_bound_consts(searchers)
    assert all(np.array_equal(s.bound_consts, searchers[0].bound_consts) for s in searchers)
    runs = [DeviceSearch(s, 32, B, k, 2) for s in searchers]
    tops = torch.stack([r.phase1(Qdev).clone() for r in runs])
    torch.cuda.synchronize()
    packed = []
    for r in runs:
        r.phase2(Qdev, tops)
        torch.cuda.synchronize()
        packed.append(r.packed.clone())
    mp, ms = merge_packed(torch.stack(packed), B, k)
    torch.cuda.synchronize()
    mp = mp.cpu().numpy(); ms = ms.cpu().numpy()
    for j in range(B):
        rp, rs, _ = oracle.search(idx, Qs[:, :, j], 2, k)
        assert np.array_equal(mp[j], rp), j
        same_f32(ms[j], rs, "heterogeneous shards")
    # phase 2 without its phase 1 is refused
    with pytest.raises(clb.ArgumentError):
        runs[0].phase2(Qdev, tops)
    for s in searchers:
        s.close()


# ---------------------------------------------------------------------------------------------------
# config 4: 1 M passages
# ---------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def million():
    idx = synthetic.make_index(seed=2024, n_docs=1_000_000, n_blocks=8)          # bench.py's corpus
    assert idx["centroids"].shape[1] == 131072
    Qs = synthetic.make_topic_queries(idx["centroids"], seed=77, n_queries=8)
    return idx, Qs


def test_config4_one_million_passages(million, oracle):
    idx, Qs = million
    emb2pid_ref = {}
    s = clb.Searcher(index=idx)
    try:
        for j in (0, 3):
            rp, rs, rn = oracle.search(idx, Qs[:, :, j], 2, 1000)
            emb2pid_ref[j] = (rp, rs)
            for mode in (0, 1):
                s.set_mode(mode)
                pids, scores = s.search_embeddings(Qs[:, :, j], 1000)
                assert s.last_num_candidates == rn
                assert np.array_equal(pids, rp), (mode, j)
                same_f32(scores, rs, f"1M scores mode={mode} q={j}")
        s.set_mode(1)
        bp, bs, _ = s.search_batch(Qs, 1000)
        for j, (rp, rs) in emb2pid_ref.items():
            assert np.array_equal(bp[:, j], rp)
            same_f32(bs[:, j], rs, "1M batch")
    finally:
        s.close()


def test_config4_eight_shards_two_phase(million, oracle):
    """The 8-way sharded index of config 4 on ONE GPU: shards searched one after the other with the two-phase
    protocol (stacking the per-shard score blocks stands in for the all-gather), merged == oracle on the full index."""
    torch = pytest.importorskip("torch")
    from colbert_jl_amd.distributed import DeviceSearch, merge_packed, share_bound_consts
    idx, Qs = million
    k, B, world = 1000, Qs.shape[2], 8
    Qdev = torch.from_numpy(np.ascontiguousarray(Qs.transpose(2, 1, 0))).cuda()
    searchers = []
    for r in range(world):
        sh = synthetic.make_index(seed=2024, n_docs=1_000_000, n_blocks=8, blocks=[r])
        searchers.append(clb.Searcher(index=sh, pid_offset=int(sh["pid_offset"])))
        del sh
    share_bound_consts(searchers)
    runs = [DeviceSearch(s, 32, B, k, 2) for s in searchers]
    tops = torch.stack([r.phase1(Qdev).clone() for r in runs])
    torch.cuda.synchronize()
    packed = []
    for r in runs:
        r.phase2(Qdev, tops)
        torch.cuda.synchronize()
        packed.append(r.packed.clone())
    mp, ms = merge_packed(torch.stack(packed), B, k)
    torch.cuda.synchronize()
    mp = mp.cpu().numpy(); ms = ms.cpu().numpy()
    for j in (0, 3, 7):
        rp, rs, _ = oracle.search(idx, Qs[:, :, j], 2, k)
        assert np.array_equal(mp[j], rp), j
        same_f32(ms[j], rs, "8-shard two-phase")
    for s in searchers:
        s.close()


# ---------------------------------------------------------------------------------------------------
# config 2: the index build at 100 k passages
# ---------------------------------------------------------------------------------------------------
def test_config2_index_build_100k(oracle):
    embs, doclens = synthetic.make_embeddings(seed=61, n_docs=100_000)
    n_emb = embs.shape[1]
    rng = np.random.default_rng(62)
    off = np.concatenate([[0], np.cumsum(doclens)])
    pids = np.unique(rng.integers(0, 100_000, size=codec.num_sampled_pids(100_000)))
    cols = np.concatenate([np.arange(off[p], off[p + 1]) for p in pids])
    sample = np.asfortranarray(embs[:, rng.permutation(cols)])
    h = codec.heldout_size(sample.shape[1])
    sample, held = sample[:, :-h], sample[:, -h:]
    plan = codec.setup(100_000, float(doclens[pids].mean()), sample.shape[1], 25000, 1)
    K = plan["num_partitions"]
    assert K == 32768
    init = sample[:, rng.permutation(sample.shape[1])[:K]]
    cent, assign, it = codec.kmeans(sample, init, max_iters=4)
    assert it == 4 and assign.min() >= 1 and assign.max() <= K
    # the assignment of the last iteration is the nearest centroid of the PREVIOUS centroids; check the update rule
    # on a sample of clusters instead: a centroid is the mean of its members (fp32, ordered sum)
    cut, w, avg, _ = codec.compute_avg_residuals(2, cent, held)
    rcut, rw, ravg, _ = oracle.compute_avg_residuals(2, cent, held)
    same_f32(cut, rcut, "cutoffs at 100k"); same_f32(w, rw, "weights at 100k")
    assert np.isclose(avg, ravg, rtol=1e-5)
    chunk = 2_000_000
    parts = [codec.compress(cent, cut, 128, 2, embs[:, i:i + chunk]) for i in range(0, n_emb, chunk)]
    codes = np.concatenate([p[0] for p in parts]); res = np.concatenate([p[1] for p in parts], axis=1)
    sel = np.sort(rng.choice(n_emb, size=20_000, replace=False))
    rc, rr = oracle.compress(cent, cut, 128, 2, np.asfortranarray(embs[:, sel]))
    assert np.array_equal(codes[sel], rc) and np.array_equal(res[:, sel], rr)
    ivf, lens = codec.build_ivf(codes, K)
    rivf, rlens = oracle.build_ivf(codes, K)
    assert np.array_equal(ivf, rivf) and np.array_equal(lens, rlens)


# ---------------------------------------------------------------------------------------------------
# config 5, index-build half: ONE rank's 1/8 share of the 10 M-passage build (K = 262 144, ~5.5 M of the 44 M sample
# points, 1.25 M passages = 100 M embeddings to compress) through the sharded product path over RCCL (world 1)
# ---------------------------------------------------------------------------------------------------
def test_config5_one_rank_share_of_the_index_build(oracle):
    import os
    import tempfile

    import torch
    import torch.distributed as dist
    from colbert_jl_amd.distributed_index import HipBackend, index_device_sharded
    free, _total = torch.cuda.mem_get_info(0)
    if free < 60e9:
        pytest.skip("needs ~40 GB of HBM")
    dev = torch.device("cuda", 0)
    n_total, n_local = 10_000_000, 1_250_000
    src = synthetic.DeviceMixtureSource(seed=71, n_docs=n_local, device=dev)
    store = tempfile.NamedTemporaryFile(prefix="clb_pg_", delete=False); store.close(); os.unlink(store.name)
    dist.init_process_group("nccl", init_method="file://" + store.name, rank=0, world_size=1, device_id=dev)
    keep = {}
    try:
        index, rec = index_device_sharded(src, 0, n_total, HipBackend(0), nbits=2, kmeans_niters=2, seed=72, keep=keep)
    finally:
        dist.destroy_process_group()
    K = rec["K"]
    assert K == 262144 and rec["kmeans_iters"] == 2 and 5_000_000 < rec["sample_points_local"] < 6_000_000
    sample, init = keep["sample"], keep["init"]
    init_t = torch.from_numpy(np.ascontiguousarray(init.T)).to(dev)
    # (1) the exchange over RCCL changes nothing: the same two iterations without a process group, bit for bit
    c1, _, sh1 = codec.kmeans_device(sample, init_t, max_iters=1)
    sh1.close()
    c2, it2, sh2 = codec.kmeans_device(sample, init_t, max_iters=2)
    assign = sh2.get_assignments()                      # iteration 2 assigned against c1
    sh2.close()
    assert it2 == 2 and torch.equal(c2, index["centroids"])
    # (2) assignments at K = 262 144 on a sampled subset == the oracle's nearest centroid (k-means distance) against c1
    rng = np.random.default_rng(73)
    sel = np.sort(rng.choice(sample.shape[0], size=1500, replace=False))
    c1_host = np.asfortranarray(c1.cpu().numpy().T)
    pts = np.asfortranarray(sample[torch.from_numpy(sel).to(dev)].cpu().numpy().T)
    _s, _c, ra = oracle.kmeans_shard_pass(pts, c1_host)
    assert np.array_equal(assign[sel], ra)
    assert assign.min() >= 1 and assign.max() <= K
    # (3) the update rule (utils.jl:288-306) on sampled clusters, restated: per batch of 1000 points a partial sum of the
    # members in ascending order, partials added in batch order, divided by max(count, 1); empty clusters -> 0
    c2_host = c2.cpu().numpy()
    order = np.argsort(assign, kind="stable")
    starts = np.searchsorted(assign[order], np.arange(1, K + 2))
    for j in rng.choice(K, size=60, replace=False):
        mem = order[starts[j]:starts[j + 1]]
        x = sample[torch.from_numpy(mem).to(dev)].cpu().numpy() if mem.size else np.zeros((0, 128), np.float32)
        total = np.zeros(128, dtype=np.float32)
        for b in np.unique(mem // 1000):
            part = np.zeros(128, dtype=np.float32)
            for r in np.nonzero(mem // 1000 == b)[0]:
                part = part + x[r]
            total = total + part
        want = total / np.float32(max(mem.size, 1))
        same_f32(c2_host[j], want, f"centroid {j} ({mem.size} members)")
    # (4) compress at K = 262 144: sampled embeddings of sampled chunks == oracle.compress, bytes and codes
    host_c = np.asfortranarray(c2_host.T)
    cut = index["bucket_cutoffs"]
    off = np.concatenate([[0], np.cumsum(src.doclens)])
    codes = index["codes"]
    for start in (0, 600_000, 1_225_000):
        x = src.chunk(start, start + 25_000)
        pick = np.sort(rng.choice(x.shape[0], size=400, replace=False))
        xs = np.asfortranarray(x[torch.from_numpy(pick).to(dev)].cpu().numpy().T)
        rc, rr = oracle.compress(host_c, cut, 128, 2, xs)
        gidx = torch.from_numpy(pick + off[start]).to(dev)
        assert np.array_equal(codes[gidx].cpu().numpy().view(np.uint32), rc)
        assert np.array_equal(index["residuals"][gidx].cpu().numpy().T, rr)
    # (5) the IVF over all 100 M codes == the oracle's stable sort
    hc = codes.cpu().numpy().view(np.uint32)
    rivf, rlens = oracle.build_ivf(hc, K)
    assert np.array_equal(index["ivf"].cpu().numpy(), rivf) and np.array_equal(index["ivf_lengths"], rlens)


# ---------------------------------------------------------------------------------------------------
# config 5: 10 M passages on one GPU (properties only; the oracle would need minutes per query)
# ---------------------------------------------------------------------------------------------------
def test_config5_ten_million_properties(oracle):
    psutil = pytest.importorskip("psutil")
    if psutil.virtual_memory().available < 150e9:
        pytest.skip("needs ~100 GB of host memory to generate the 10 M-passage index")
    n_docs, n_blocks = 10_000_000, 8
    K = synthetic.num_partitions_for(n_docs, 80.0)
    assert K == 262144
    idx = synthetic.make_index(seed=2024, n_docs=n_docs, K=K, n_blocks=n_blocks, ivf_on_device=True)
    Qs = synthetic.make_topic_queries(idx["centroids"], seed=77, n_queries=4)
    s = clb.Searcher(index=idx)
    try:
        k = 1000
        out = {}
        for mode in (0, 1):
            s.set_mode(mode)
            pids, scores, n = s.search_batch(np.asfortranarray(Qs[:, :, :2]), k)
            assert np.all(np.diff(scores, axis=0) <= 0)                               # sorted
            ties = np.diff(scores, axis=0) == 0
            assert np.all(np.diff(pids, axis=0)[ties] > 0)                            # ties by ascending pid
            for j in range(2):
                assert np.unique(pids[:, j]).size == k and pids[:, j].min() >= 1 and pids[:, j].max() <= n_docs
            out[mode] = (pids, scores)
        assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(bits(out[0][1]), bits(out[1][1]))
        s.set_mode(1)
        p4, s4, _ = s.search_batch(Qs, k)                                             # a different batch size
        assert np.array_equal(p4[:, :2], out[1][0]) and np.array_equal(bits(s4[:, :2]), bits(out[1][1]))
        # ... and the oracle itself on two of the queries (one of them only seen in the batch of four): same pids, same bits
        idx["emb2pid"] = oracle.build_emb2pid(idx["doclens"])
        for j in (0, 3):
            rp, rs, _ = oracle.search(idx, Qs[:, :, j], 2, k)
            assert np.array_equal(p4[:, j], rp) and np.array_equal(bits(s4[:, j]), bits(rs)), j
    finally:
        s.close()
