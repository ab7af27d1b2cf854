"""Multi-GPU index build (BASELINE config 5): one process per GPU, the passages -- hence the sampled
points of the k-means -- sharded over the ranks.

  * k-means (`kmeans_gpu_onehot!`, src/utils.jl:253-318): every rank keeps its points on its device; per iteration it
    computes the un-normalised cluster sums and counts of its points (clb_kmeans_shard_pass), one all-gather moves the
    (K*dim + K)-sized blocks (RCCL over xGMI; 64 MB per rank at K = 131 072 -- the one bandwidth-relevant exchange of
    the build), and every rank reduces them in RANK ORDER and updates the centroids (clb_kmeans_reduce_update):
    deterministic and identical on all ranks.  The association of the fp32 sums differs from the single-device loop
    (sum over all batches in order) only across shard boundaries: results agree to fp32 rounding, and bit for bit
    with the sharded restatement in the oracle.
  * codec statistics (`_compute_avg_residuals!`, collection_indexer.jl:177-195): <= 50 000 held-out embeddings -- rank
    0 computes them, the result (2^nbits - 1 cutoffs, 2^nbits weights, avg_residual) is broadcast.
  * compress + IVF: per embedding / per shard, no exchange: each rank compresses its passages and builds the IVF of its
    shard, which is exactly what its `Searcher` shard needs (sharding.py).

The compute callbacks are injected (`backend`): the HIP library in production, the CPU oracle in the gloo tests."""
from __future__ import annotations

import numpy as np


class HipBackend:
    """The product path: libcolbert_hip through codec.py."""

    def __init__(self, device: int = 0):
        self.device = device

    def shard(self, data, K, point_bsize):
        from . import codec
        sh = codec.KMeansShard(data, K, point_bsize, device=self.device)
        return sh.pass_

    def device_shard(self, data, K, point_bsize):
        """The shard handle itself: kmeans_sharded then keeps centroids, partial sums and the exchange on the device."""
        from . import codec
        return codec.KMeansShard(data, K, point_bsize, device=self.device)

    def reduce_update(self, centroids, gs, gc, tol):
        from . import codec
        return codec.kmeans_reduce_update(centroids, gs, gc, tol, device=self.device)

    def compute_avg_residuals(self, nbits, centroids, heldout):
        from . import codec
        return codec.compute_avg_residuals(nbits, centroids, heldout, device=self.device)[:3]

    def compress(self, centroids, cutoffs, dim, nbits, embs):
        from . import codec
        return codec.compress(centroids, cutoffs, dim, nbits, embs, device=self.device)

    def build_ivf(self, codes, K):
        from . import codec
        return codec.build_ivf(codes, K, device=self.device)


def _all_gather(t, group):
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    out = torch.empty((world,) + tuple(t.shape), dtype=t.dtype, device=t.device)
    dist.all_gather_into_tensor(out.view(world * t.shape[0], *t.shape[1:]) if t.dim() > 1 else out.view(-1), t.contiguous(),
                                group=group)
    return out


def kmeans_sharded(local_points, init_centroids, backend, max_iters: int = 10, tol: float = 1e-4, point_bsize: int = 1000,
                   group=None, comm_device=None, all_gather=None):
    """kmeans_gpu_onehot! over points sharded across the ranks of `group` (contiguous, rank order).
    local_points (dim, n_local) fp32; init_centroids (dim, K), identical on every rank.
    Returns (centroids, iterations executed) -- identical on every rank."""
    import torch
    import torch.distributed as dist
    c = np.asfortranarray(init_centroids, dtype=np.float32).copy(order="F")
    dim, K = c.shape
    if comm_device is not None and comm_device.type == "cuda" and hasattr(backend, "device_shard"):
        # product path: centroids stay in the shard handle, every rank's [sums | counts] block goes straight from the
        # accumulation kernels into the all-gather (RCCL) and from there into the reduction kernel -- no numpy round trip
        sh = backend.device_shard(local_points, K, point_bsize)
        world = dist.get_world_size(group)
        sh.set_centroids(c)
        blk = torch.empty(sh.block_bytes, dtype=torch.uint8, device=comm_device)
        gathered = torch.empty(world * sh.block_bytes, dtype=torch.uint8, device=comm_device)
        it = 0
        with torch.cuda.device(comm_device):
            for it in range(1, max_iters + 1):
                sh.pass_device(blk)
                if all_gather is not None:
                    gathered = all_gather(blk).reshape(-1)          # e.g. LibraryComm.all_gather (clb_comm_all_gather)
                else:
                    dist.all_gather_into_tensor(gathered, blk, group=group)
                _delta, conv = sh.update_device(gathered, world, tol)
                if conv:
                    break
        c = sh.get_centroids()
        sh.close()
        return c, (it if max_iters > 0 else 0)
    pass_ = backend.shard(local_points, K, point_bsize)
    dev = comm_device if comm_device is not None else torch.device("cpu")
    it = 0
    for it in range(1, max_iters + 1):
        sums, counts = pass_(c)
        blk = torch.from_numpy(np.concatenate([sums.ravel(order="F").view(np.int32), counts.astype(np.int64).view(np.int32)]))
        g = _all_gather(blk.to(dev), group).cpu().numpy()                      # (world, dim*K + 2K) int32 words
        gs = g[:, :dim * K].copy().view(np.float32)
        gc = np.ascontiguousarray(g[:, dim * K:]).view(np.int64)
        c, _delta, conv = backend.reduce_update(c, gs, gc, tol)
        if conv:
            break
    if max_iters <= 0:
        it = 0
    return c, it


def build_index_sharded(local_embs, local_doclens, local_sample, heldout, init_centroids, backend, nbits: int = 2,
                        kmeans_niters: int = 20, group=None, comm_device=None):
    """The array stages of index() (src/indexing.jl:102-143) for one passage shard.
    local_embs (dim, n_local): this rank's passage embeddings; local_sample: its part of the clustering sample;
    heldout: the held-out embeddings (used on rank 0 only); init_centroids: identical on every rank.
    Returns the fields a `Searcher` shard needs."""
    import torch
    import torch.distributed as dist
    rank = dist.get_rank(group)
    dim = local_embs.shape[0]
    centroids, iters = kmeans_sharded(local_sample, init_centroids, backend, max_iters=kmeans_niters, group=group,
                                      comm_device=comm_device)
    K = centroids.shape[1]
    nopt = 1 << nbits
    stats = np.zeros(2 * nopt, dtype=np.float32)                  # cutoffs (nopt-1), weights (nopt), avg_residual
    if rank == 0:
        cut, w, avg = backend.compute_avg_residuals(nbits, centroids, heldout)
        stats[:nopt - 1] = cut; stats[nopt - 1:2 * nopt - 1] = w; stats[2 * nopt - 1] = avg
    dev = comm_device if comm_device is not None else torch.device("cpu")
    t = torch.from_numpy(stats).to(dev)
    dist.broadcast(t, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
    stats = t.cpu().numpy()
    cutoffs, weights, avg = stats[:nopt - 1].copy(), stats[nopt - 1:2 * nopt - 1].copy(), np.float32(stats[2 * nopt - 1])
    codes, residuals = backend.compress(centroids, cutoffs, dim, nbits, local_embs)
    ivf, ivf_lengths = backend.build_ivf(codes, K)
    return {"dim": dim, "nbits": nbits, "centroids": centroids, "bucket_cutoffs": cutoffs, "bucket_weights": weights,
            "avg_residual": avg, "doclens": np.asarray(local_doclens, dtype=np.int64), "codes": codes,
            "residuals": residuals, "ivf": ivf, "ivf_lengths": ivf_lengths, "kmeans_iters": iters}


def index_device_sharded(source, pid_lo: int, n_docs_total: int, backend=None, nbits: int = 2, kmeans_niters: int = 20,
                         seed: int = 0, group=None, all_gather=None, chunksize=None, log=None, keep=None):
    """One rank's part of index() (src/indexing.jl:63-147) for a collection sharded by contiguous passage ranges, with
    every large array resident in HBM (BASELINE config 5: the fp32 embeddings of a 1.25 M-passage shard are 51 GB and
    exist only chunk by chunk).  `source` (indexer.DeviceEmbeddingSource) yields THIS rank's passages -- local ids
    0..n_local-1 are the global passages pid_lo..pid_lo+n_local-1 of `n_docs_total`.

      sample   the reference's rule over the WHOLE collection (collection_indexer.jl:17-24), drawn identically on every
               rank; a rank gathers the sampled passages of its own range.  Shuffled per rank; the held-out set
               (collection_indexer.jl:81-91: min(5 %, 50 000) of all sample embeddings) is the tail of rank 0's part.
      setup    K from the global estimate (all-reduce of the sample counts), collection_indexer.jl:115-139.
      init     every rank contributes ceil(K / world) of its sample points, all-gathered in rank order (utils.jl:261
               draws K random points of the sample).
      k-means  kmeans_sharded: shard handle over the rank's device points, one all-gather per iteration.
      stats    rank 0, broadcast.   compress + IVF: per shard (resident codec, device IVF), no exchange.

    Returns (index, record): `index` is what Searcher(index=..., pid_offset=pid_lo) takes (CUDA tensors + host lengths).
    `keep` (a dict, tests only) receives this rank's clustering sample and the initial centroids."""
    import time

    import torch
    import torch.distributed as dist

    from . import codec
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    backend = backend or HipBackend(source.device.index)
    dev = source.device
    doclens = np.ascontiguousarray(source.doclens, dtype=np.int64)
    n_local, dim = doclens.size, source.dim
    off = np.concatenate([[0], np.cumsum(doclens)])
    n_emb = int(off[-1])
    rec = {"rank": rank, "world": world, "passages_total": int(n_docs_total), "passages_local": int(n_local),
           "embeddings_local": n_emb}

    def tick(name, t0):
        torch.cuda.synchronize(dev)
        rec[name] = round(time.time() - t0, 3)
        if log:
            log(f"index_device_sharded[{rank}]: {name} {rec[name]} s")

    # RCCL moves device tensors; under gloo (tests: several ranks on one GPU, CPU rehearsals) the same exchanges are
    # staged through host memory
    staged = dist.get_backend(group) != "nccl"
    cdev = torch.device("cpu") if staged else dev

    def all_sum(vals):
        t = torch.tensor(vals, dtype=torch.int64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        return [int(v) for v in t.tolist()]

    def gather_rows(t):
        """contiguous (n, ...) device tensor -> (world, n, ...) device tensor, rank order"""
        src_t = t.contiguous().to(cdev)
        out = torch.empty((world,) + tuple(src_t.shape), dtype=src_t.dtype, device=cdev)
        dist.all_gather_into_tensor(out.view(-1), src_t.view(-1), group=group)
        return out.to(dev)

    if all_gather is None and staged:
        all_gather = gather_rows

    # sample
    t0 = time.time()
    rng = np.random.default_rng(seed)
    sampled = np.unique(rng.integers(0, n_docs_total, size=codec.num_sampled_pids(n_docs_total)))
    mine = sampled[(sampled >= pid_lo) & (sampled < pid_lo + n_local)] - pid_lo
    n_sample = int(doclens[mine].sum())
    # a source that can produce the embeddings of a list of passages (the encoder: indexer.EncoderSource) encodes only the
    # sample, as the reference does (collection_indexer.jl:56-79); else they are cut out of the chunks they fall in
    from .indexer import _sample_from_chunks
    step = int(chunksize or min(25000, 1 + n_docs_total // world))
    sample = source.sample(mine) if hasattr(source, "sample") else _sample_from_chunks(source, mine, step)
    fill = int(sample.shape[0])
    assert fill == n_sample
    lrng = np.random.default_rng([seed, rank + 1])
    sample = codec.gather_rows_device(sample.contiguous(), lrng.permutation(n_sample))      # the shuffle, through the C ABI
    tot_sample, tot_sampled_docs = all_sum([n_sample, int(mine.size)])
    h = codec.heldout_size(tot_sample)
    heldout = None
    if rank == 0:
        assert n_sample > h, "rank 0 holds fewer sample embeddings than the held-out set"
        heldout = sample[n_sample - h:]
        sample = sample[:n_sample - h]
    avg_doclen_est = float(np.float32(tot_sample / max(tot_sampled_docs, 1)))
    plan = codec.setup(n_docs_total, avg_doclen_est, tot_sample - h, chunksize, world)
    K = plan["num_partitions"]
    per = -(-K // world)
    assert sample.shape[0] >= per, "a rank holds fewer sample points than its share of the initial centroids"
    mine_init = sample[:per].contiguous()
    allinit = gather_rows(mine_init).view(world * per, dim) if world > 1 else mine_init
    init = np.asfortranarray(allinit[:K].cpu().numpy().T)
    sample = sample.contiguous()
    rec.update({"sample_points_local": int(sample.shape[0]), "sample_points_total": int(tot_sample - h), "heldout": int(h),
                "K": int(K), "chunksize": plan["chunksize"]})
    tick("sample_and_split_s", t0)
    if keep is not None:
        keep["sample"], keep["init"], keep["heldout"] = sample, init, heldout

    # k-means over the process group
    t0 = time.time()
    centroids, iters = kmeans_sharded(sample, init, backend, max_iters=kmeans_niters, group=group, comm_device=dev,
                                      all_gather=all_gather)
    tick("kmeans_s", t0)
    rec["kmeans_iters"] = int(iters)
    rec["kmeans_s_per_iter"] = round(rec["kmeans_s"] / max(iters, 1), 4)
    rec["kmeans_exchange_bytes_per_rank_per_iter"] = int(dim * K * 4 + K * 8)
    t0 = time.time()
    nopt = 1 << nbits
    stats = np.zeros(2 * nopt, dtype=np.float32)
    if rank == 0:
        cut, w, avg = backend.compute_avg_residuals(nbits, centroids, np.asfortranarray(heldout.cpu().numpy().T))
        stats[:nopt - 1] = cut; stats[nopt - 1:2 * nopt - 1] = w; stats[2 * nopt - 1] = avg
    t = torch.from_numpy(stats).to(cdev)
    dist.broadcast(t, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
    stats = t.cpu().numpy()
    cutoffs, weights, avg = stats[:nopt - 1].copy(), stats[nopt - 1:2 * nopt - 1].copy(), np.float32(stats[2 * nopt - 1])
    tick("codec_stats_s", t0)
    del sample, heldout

    # this shard's chunks
    t0 = time.time()
    rows_b = dim // 8 * nbits
    codes = torch.empty(n_emb, dtype=torch.int32, device=dev)
    residuals = torch.empty((n_emb, rows_b), dtype=torch.uint8, device=dev)
    cdc = codec.Codec(centroids, cutoffs, dim, nbits, device=dev.index)
    for start in range(0, n_local, plan["chunksize"]):
        end = min(n_local, start + plan["chunksize"])
        x = source.chunk(start, end)
        a, b = int(off[start]), int(off[end])
        cdc.compress_device(x, codes[a:b], residuals[a:b])
        torch.cuda.synchronize(dev)
        del x
    cdc.close()
    tick("chunks_s", t0)
    rec["compress_Membeddings_per_s"] = round(n_emb / max(rec["chunks_s"], 1e-9) / 1e6, 2)
    t0 = time.time()
    ivf, ivf_lengths = codec.build_ivf_device(codes, K)
    tick("build_ivf_s", t0)
    rec["total_build_s"] = round(sum(rec[k] for k in ("sample_and_split_s", "kmeans_s", "codec_stats_s", "chunks_s", "build_ivf_s")), 2)
    cent_dev = torch.from_numpy(np.ascontiguousarray(centroids.T)).to(dev)
    index = {"dim": dim, "nbits": nbits, "centroids": cent_dev, "bucket_cutoffs": cutoffs, "bucket_weights": weights,
             "avg_residual": avg, "doclens": doclens, "codes": codes, "residuals": residuals, "ivf": ivf,
             "ivf_lengths": ivf_lengths.cpu().numpy(), "pid_offset": int(pid_lo), "kmeans_iters": int(iters)}
    return index, rec
