"""Multi-GPU index build, orchestration on CPU: world_size 2 over gloo with the CPU oracle as the compute backend
(the HIP library has no CPU fallback; the same orchestration runs over libcolbert_hip -- distributed_index.HipBackend -- in
tests/test_gpu_dist_index.py: device-resident exchange, RCCL with one rank, two rank processes on one GPU).
Stage-wise parity: the distributed k-means equals the oracle's sharded restatement bit for bit, a single shard equals
the reference loop bit for bit, the 2-shard centroids agree with the single-device loop to fp32 rounding, and the
per-shard codes / residual bytes / IVF equal the oracle's on the same inputs."""
import os
import socket

import numpy as np
import pytest

import colbert_jl_amd as clb  # noqa: F401
from colbert_jl_amd import synthetic


class OracleBackend:
    def shard(self, data, K, point_bsize):
        from oracle import oracle as orc
        data = np.asfortranarray(data, dtype=np.float32)
        return lambda c: orc.kmeans_shard_pass(data, c, point_bsize)[:2]

    def reduce_update(self, centroids, gs, gc, tol):
        from oracle import oracle as orc
        return orc.kmeans_reduce_update(centroids, gs, gc, tol)

    def compute_avg_residuals(self, nbits, centroids, heldout):
        from oracle import oracle as orc
        return orc.compute_avg_residuals(nbits, centroids, heldout)[:3]

    def compress(self, centroids, cutoffs, dim, nbits, embs):
        from oracle import oracle as orc
        return orc.compress(centroids, cutoffs, dim, nbits, embs)

    def build_ivf(self, codes, K):
        from oracle import oracle as orc
        return orc.build_ivf(codes, K)


def _problem(world=2):
    """120 passages cut into `world` contiguous ranges; every rank samples columns of its own range."""
    embs, doclens = synthetic.make_embeddings(seed=71, n_docs=120, dim=32, doclen_mean=20, doclen_std=4, n_components=12)
    rng = np.random.default_rng(72)
    per = 120 // world
    off = np.concatenate([[0], np.cumsum(doclens)])
    bounds = [(int(off[r * per]), int(off[(r + 1) * per] if r + 1 < world else off[-1])) for r in range(world)]
    n_s = 1350 // world
    sample_cols = [np.sort(lo + rng.choice(hi - lo, min(n_s, hi - lo), replace=False)) for lo, hi in bounds]
    held = np.asfortranarray(embs[:, rng.choice(embs.shape[1], 200, replace=False)])
    K = 24
    init = np.asfortranarray(embs[:, rng.choice(embs.shape[1], K, replace=False)])
    return embs, doclens, bounds, sample_cols, held, init


def _worker(rank, world, port, q):
    import torch.distributed as dist
    from colbert_jl_amd.distributed_index import build_index_sharded
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    embs, doclens, bounds, sample_cols, held, init = _problem(world)
    lo, hi = bounds[rank]
    per = 120 // world
    dl = doclens[rank * per:(rank + 1) * per if rank + 1 < world else 120]
    out = build_index_sharded(np.asfortranarray(embs[:, lo:hi]), dl, np.asfortranarray(embs[:, sample_cols[rank]]),
                              held, init, OracleBackend(), nbits=2, kmeans_niters=6)
    q.put((rank, {k: v for k, v in out.items()}))
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_sharded_index_build_over_gloo(oracle, world):
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=240) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    embs, doclens, bounds, sample_cols, held, init = _problem(world)
    # (1) identical centroids and codec statistics on every rank
    for r in range(1, world):
        for k in ("centroids", "bucket_cutoffs", "bucket_weights"):
            assert np.array_equal(res[0][k].view(np.uint32), res[r][k].view(np.uint32)), (k, r)
    # (2) == the oracle's sharded restatement, run in one process (partial sums added in rank order)
    c = init.copy(order="F")
    shards = [np.asfortranarray(embs[:, cols]) for cols in sample_cols]
    iters = 0
    for iters in range(1, 7):
        parts = [oracle.kmeans_shard_pass(x, c)[:2] for x in shards]
        c, _d, conv = oracle.kmeans_reduce_update(c, [p[0] for p in parts], [p[1] for p in parts])
        if conv:
            break
    assert res[0]["kmeans_iters"] == iters
    assert np.array_equal(res[0]["centroids"].view(np.uint32), c.view(np.uint32))
    # (3) close to the single-device loop over the concatenated sample (association differs across the shard boundaries)
    c1, _, _ = oracle.kmeans(np.asfortranarray(np.concatenate(shards, axis=1)), init, max_iters=6)
    assert np.allclose(res[0]["centroids"], c1, rtol=0, atol=2e-6)
    # (4) codec statistics == oracle on rank 0's inputs; per-shard codes / residuals / IVF == oracle
    rcut, rw, ravg, _ = oracle.compute_avg_residuals(2, c, held)
    assert np.array_equal(res[world - 1]["bucket_cutoffs"].view(np.uint32), rcut.view(np.uint32))
    assert np.array_equal(res[world - 1]["bucket_weights"].view(np.uint32), rw.view(np.uint32))
    for rank, (lo, hi) in enumerate(bounds):
        rc, rr = oracle.compress(c, rcut, 32, 2, np.asfortranarray(embs[:, lo:hi]))
        assert np.array_equal(res[rank]["codes"], rc) and np.array_equal(res[rank]["residuals"], rr)
        rivf, rlen = oracle.build_ivf(rc, 24)
        assert np.array_equal(res[rank]["ivf"], rivf) and np.array_equal(res[rank]["ivf_lengths"], rlen)
    # the shards concatenate to the unsharded compression
    rc_all, _ = oracle.compress(c, rcut, 32, 2, embs)
    assert np.array_equal(np.concatenate([res[r]["codes"] for r in range(world)]), rc_all)


def test_single_shard_equals_reference_loop(oracle):
    rng = np.random.default_rng(5)
    X = np.asfortranarray(rng.standard_normal((16, 2300)).astype(np.float32))
    C0 = np.asfortranarray(X[:, :17].copy())
    c = C0.copy(order="F")
    for _ in range(3):
        s, n, _a = oracle.kmeans_shard_pass(X, c, 1000)
        c, _d, conv = oracle.kmeans_reduce_update(c, [s], [n])
        assert not conv
    ref, _, it = oracle.kmeans(X, C0, max_iters=3, point_bsize=1000)
    assert it == 3 and np.array_equal(ref.view(np.uint32), c.view(np.uint32))
