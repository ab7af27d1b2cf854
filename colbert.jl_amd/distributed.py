"""Multi-GPU search: one process per GPU, passages sharded across ranks (sharding.py), every rank runs
the whole search on its shard and one all-gather of the per-shard top-k (k (pid, score) records per
query, ~12 KB per rank at k = 1000) is the only exchange -- RCCL over xGMI through torch.distributed
(backend "nccl" is RCCL on ROCm).  The merged list is identical to the unsharded result because
candidates and scores are per passage.

On CPU (gloo, used by the tests) the same code runs with a caller-supplied local search function and a
host merge."""
from __future__ import annotations

import ctypes as C

import numpy as np

from ._lib import check, i64, lib
from .sharding import merge_topk_host


def all_gather_topk(pids, scores, group=None):
    """pids/scores: torch tensors (B, k) on this rank -> stacked (world, B, k) tensors on every rank."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    B, k = pids.shape
    gp = torch.empty((world * B, k), dtype=pids.dtype, device=pids.device)      # concatenation along dim 0
    gs = torch.empty((world * B, k), dtype=scores.dtype, device=scores.device)
    dist.all_gather_into_tensor(gp, pids.contiguous(), group=group)
    dist.all_gather_into_tensor(gs, scores.contiguous(), group=group)
    return gp.view(world, B, k), gs.view(world, B, k)


def merge_gathered(gp, gs, k: int, device_index=None, out_p=None, out_s=None):
    """(world, B, k) gathered records -> (B, k) merged top-k.  GPU tensors go through the HIP merge
    kernel (clb_merge_topk_device) on torch's current stream; CPU tensors through the host merge."""
    import torch
    world, B, _ = gp.shape
    if gp.is_cuda:
        if out_p is None:
            out_p = torch.empty((B, k), dtype=torch.int64, device=gp.device)
            out_s = torch.empty((B, k), dtype=torch.float32, device=gp.device)
        st = torch.cuda.current_stream(gp.device).cuda_stream
        check(lib().clb_merge_topk_device(gp.device.index if device_index is None else device_index,
                                          C.c_void_p(gp.data_ptr()), C.c_void_p(gs.data_ptr()), i64(k), i64(world),
                                          i64(B), C.c_void_p(out_p.data_ptr()), C.c_void_p(out_s.data_ptr()),
                                          C.c_void_p(st)))
        return out_p, out_s
    P = gp.numpy(); S = gs.numpy()
    out_p = np.zeros((B, k), dtype=np.int64); out_s = np.full((B, k), -np.inf, dtype=np.float32)
    for b in range(B):
        mp, ms = merge_topk_host([P[r, b] for r in range(world)], [S[r, b] for r in range(world)], k)
        out_p[b, : mp.size] = mp; out_s[b, : ms.size] = ms
    return torch.from_numpy(out_p), torch.from_numpy(out_s)


def sharded_search(local_search, Q, k: int, group=None):
    """local_search(Q) -> (pids (B, k), scores (B, k)) torch tensors for this rank's shard, padded with
    (0, -inf).  Returns the merged (B, k) result, identical on every rank."""
    p, s = local_search(Q)
    gp, gs = all_gather_topk(p, s, group)
    return merge_gathered(gp, gs, k)


class DeviceSearch:
    """Device-resident batched search on one shard: queries and results stay in HBM (torch tensors),
    work is enqueued on torch's current stream."""

    def __init__(self, searcher, T: int, B: int, k: int, nprobe: int):
        import torch
        self.s, self.T, self.B, self.k, self.nprobe = searcher, T, B, k, nprobe
        self.dev = torch.device("cuda", searcher.device)
        self.out_p = torch.empty((B, k), dtype=torch.int64, device=self.dev)
        self.out_s = torch.empty((B, k), dtype=torch.float32, device=self.dev)
        self.ncand = torch.zeros(B, dtype=torch.int64, device=self.dev)

    def __call__(self, Qdev):
        """Qdev: torch float32 tensor holding B queries laid out (B, T, dim) contiguous == Julia (dim, T, B)."""
        import torch
        st = torch.cuda.current_stream(self.dev).cuda_stream
        check(lib().clb_search_batch_device(self.s._h, C.c_void_p(Qdev.data_ptr()), i64(self.T), i64(self.B),
                                            i64(self.nprobe), i64(self.k), C.c_void_p(self.out_p.data_ptr()),
                                            C.c_void_p(self.out_s.data_ptr()), C.c_void_p(self.ncand.data_ptr()),
                                            C.c_void_p(st)))
        return self.out_p, self.out_s
