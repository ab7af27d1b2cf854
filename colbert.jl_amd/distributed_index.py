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
