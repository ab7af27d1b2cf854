"""world_size 2 / 4 / 8 tests of the sharded search path on CPU (gloo): every rank holds a contiguous passage shard,
runs a local search, one all-gather of the per-shard top-k and the merge reproduce the unsharded result.
No GPU here, so the local search is the CPU oracle standing in for the HIP searcher (tests may use the
oracle as a checker/stand-in; the product path never does) -- what is under test is the host logic of
colbert_jl_amd.distributed / .sharding: shard bounds, pid offsets, padding of short shards, gather layout,
merge order."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, k, q, packed=False):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist

    import colbert_jl_amd as clb
    from colbert_jl_amd.distributed import sharded_search
    from colbert_jl_amd.sharding import shard_index
    from oracle import oracle as orc
    orc.set_num_threads(1 if world > 2 else 2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    idx = clb.synthetic.make_index(seed=41, n_docs=1200, K=128)
    Qs = clb.synthetic.make_queries(idx, 42, 3)
    sub, off = shard_index(idx, rank, world)

    def local_search(Q):
        B = Q.shape[2]
        P = np.zeros((B, k), np.int64); S = np.full((B, k), -np.inf, np.float32)
        for b in range(B):
            cand = orc.retrieve(sub["ivf"], sub["ivf_lengths"], sub["centroids"], orc.build_emb2pid(sub["doclens"]), 2, Q[:, :, b])
            kk = min(k, cand.size)                       # a shard may hold fewer than k candidates
            if kk:
                p, s, _ = orc.search(sub, Q[:, :, b], 2, kk)
                P[b, :kk] = p + off; S[b, :kk] = s
        return torch.from_numpy(P), torch.from_numpy(S)

    mp, ms = sharded_search(local_search, Qs, k, packed=packed)
    if rank == 0:
        ok = True
        for b in range(Qs.shape[2]):
            rp, rs, _ = orc.search(idx, Qs[:, :, b], 2, k)
            ok = ok and np.array_equal(mp[b].numpy(), rp) and np.array_equal(ms[b].numpy().view(np.uint32), rs.view(np.uint32))
        q.put(ok)
    dist.barrier()
    dist.destroy_process_group()


# world 4 and 8: the job shapes the driver's SCALE run launches (round-4 verdict: rehearse them before hardware does) -- short
# shards (1 200 passages over 8 ranks: most shards hold fewer than k candidates and pad), every rank merging
@pytest.mark.parametrize("world,k,packed", [(2, 50, False), (2, 400, False), (2, 401, True), (4, 400, True), (4, 50, False), (8, 300, True)])
def test_sharded_search_equals_unsharded(world, k, packed):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, k, q, packed)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=240)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True


def _worker_two_phase(rank, world, port, k, q):
    """The two-phase protocol of the sharded search over gloo: phase 1 publishes the shard's k largest scores per query,
    `all_gather_scores` moves them, the k-th largest of the union is the global threshold, phase 2 returns only the
    shard's passages at or above it, one packed all-gather + merge.  The oracle's exact scores stand in for the approximate
    ones (a bound of zero), so the merged result must be the unsharded top-k and the shards together must list exactly
    k passages per query (more only through ties at the threshold)."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist

    import colbert_jl_amd as clb
    from colbert_jl_amd.distributed import all_gather_packed, all_gather_scores, merge_packed, pack_topk
    from colbert_jl_amd.sharding import shard_index
    from oracle import oracle as orc
    orc.set_num_threads(1 if world > 2 else 2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    idx = clb.synthetic.make_index(seed=45, n_docs=1500, K=128)
    Qs = clb.synthetic.make_queries(idx, 46, 4)
    B = Qs.shape[2]
    sub, off = shard_index(idx, rank, world)
    local = []
    top = torch.full((B, k), -np.inf, dtype=torch.float32)
    for b in range(B):                                    # phase 1: every candidate of the shard, best first
        n = orc.retrieve(sub["ivf"], sub["ivf_lengths"], sub["centroids"], orc.build_emb2pid(sub["doclens"]), 2, Qs[:, :, b]).size
        p, s = (np.zeros(0, np.int64), np.zeros(0, np.float32)) if n == 0 else orc.search(sub, Qs[:, :, b], 2, n)[:2]
        local.append((p + off, s))
        top[b, :min(k, n)] = torch.from_numpy(s[:k].copy())
    gathered = all_gather_scores(top)                     # (world, B, k)
    assert gathered.shape == (world, B, k) and torch.equal(gathered[rank], top)
    tau = torch.sort(gathered.permute(1, 0, 2).reshape(B, world * k), dim=1, descending=True).values[:, k - 1]
    P = torch.zeros((B, k), dtype=torch.int64); S = torch.full((B, k), -np.inf, dtype=torch.float32)
    listed = torch.zeros(B, dtype=torch.int64)
    for b in range(B):                                    # phase 2: only what reaches the global threshold
        p, s = local[b]
        keep = s >= tau[b].item()
        kk = min(k, int(keep.sum()))
        P[b, :kk] = torch.from_numpy(p[keep][:kk]); S[b, :kk] = torch.from_numpy(s[keep][:kk])
        listed[b] = int(keep.sum())
    mp, ms = merge_packed(all_gather_packed(pack_topk(P, S)), B, k)
    dist.all_reduce(listed)
    if rank == 0:
        ok = True
        for b in range(B):
            rp, rs, _ = orc.search(idx, Qs[:, :, b], 2, k)
            ok = ok and np.array_equal(mp[b].numpy(), rp) and np.array_equal(ms[b].numpy().view(np.uint32), rs.view(np.uint32))
            ok = ok and k <= int(listed[b]) <= k + 8
        q.put(ok)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4, 8])
def test_two_phase_protocol(world):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_two_phase, args=(r, world, port, 100, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=240)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True


def test_shard_bounds_balance_embeddings():
    sys.path.insert(0, ROOT)
    import colbert_jl_amd as clb
    from colbert_jl_amd.sharding import shard_bounds, shard_index
    idx = clb.synthetic.make_index(seed=43, n_docs=5000, K=64)
    b = shard_bounds(idx["doclens"], 8)
    assert b[0] == 0 and b[-1] == 5000 and np.all(np.diff(b) > 0)
    per = [idx["doclens"][b[i]:b[i + 1]].sum() for i in range(8)]
    assert max(per) - min(per) <= 2 * idx["doclens"].max()
    tot = 0
    for r in range(8):
        sub, off = shard_index(idx, r, 8)
        assert off == b[r] and sub["ivf_lengths"].sum() == sub["codes"].size == sub["doclens"].sum()
        tot += sub["codes"].size
    assert tot == idx["codes"].size


def test_block_generated_shards_equal_the_full_index():
    """bench.py builds each rank's shard directly from its generation blocks; it must be the same passages the
    full index holds (codes, residuals, doclens), with the right pid offset and a local IVF over the same codes."""
    sys.path.insert(0, ROOT)
    import colbert_jl_amd as clb
    syn = clb.synthetic
    full = syn.make_index(seed=2024, n_docs=4001, K=256, n_blocks=8)
    off = np.concatenate([[0], np.cumsum(full["doclens"])])
    per = -(-4001 // 8)
    covered = 0
    for world in (2, 4, 8):
        nb = 8 // world
        for rank in range(world):
            sh = syn.make_index(seed=2024, n_docs=4001, K=256, n_blocks=8, blocks=range(rank * nb, (rank + 1) * nb))
            p0 = rank * nb * per
            p1 = min(4001, (rank + 1) * nb * per)
            assert sh["pid_offset"] == p0
            assert np.array_equal(sh["doclens"], full["doclens"][p0:p1])
            assert np.array_equal(sh["codes"], full["codes"][off[p0]:off[p1]])
            assert np.array_equal(sh["residuals"], full["residuals"][:, off[p0]:off[p1]])
            assert np.array_equal(sh["centroids"], full["centroids"])
            assert np.array_equal(sh["bucket_weights"], full["bucket_weights"])
            ivf, lens = syn.build_ivf(sh["codes"], 256)
            assert np.array_equal(sh["ivf"], ivf) and np.array_equal(sh["ivf_lengths"], lens)
            covered += p1 - p0 if world == 8 else 0
    assert covered == 4001
