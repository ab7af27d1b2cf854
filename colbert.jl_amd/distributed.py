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


def packed_topk_bytes(k: int, B: int) -> int:
    """Bytes of one rank's packed (B, k) result block: [B*k int64 pids][B*k fp32 scores][pad to 8]."""
    return (B * k * 12 + 7) // 8 * 8


class LibraryComm:
    """The exchange step on the library's own RCCL communicator (clb_comm_*), for hosts without torch.distributed --
    the same calls the Julia shim makes.  Rank 0 creates the unique id (`LibraryComm.unique_id()`) and hands its bytes to
    the other ranks by any means; every rank then constructs `LibraryComm(device, rank, n_ranks, id_bytes)` (blocks until
    all ranks have joined).  Tensors are torch CUDA tensors here only because the Python driver keeps its device memory
    in torch; the entry points take bare device pointers."""

    def __init__(self, device: int, rank: int, n_ranks: int, id_bytes: bytes):
        self._h = C.c_void_p()
        self.device, self.rank, self.n_ranks = device, rank, n_ranks
        buf = C.create_string_buffer(bytes(id_bytes), len(id_bytes))
        check(lib().clb_comm_create(C.c_int(device), C.c_int(rank), C.c_int(n_ranks), buf, i64(len(id_bytes)), C.byref(self._h)))

    @staticmethod
    def unique_id() -> bytes:
        n = int(lib().clb_comm_unique_id_bytes())
        buf = C.create_string_buffer(n)
        check(lib().clb_comm_unique_id(buf, i64(n)))
        return buf.raw

    def close(self):
        if getattr(self, "_h", None):
            lib().clb_comm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def all_gather(self, t):
        """contiguous CUDA tensor -> (n_ranks, *t.shape) tensor of the same dtype, on torch's current stream"""
        import torch
        if not (t.is_cuda and t.is_contiguous() and t.device.index == self.device):
            raise ValueError("all_gather needs a contiguous CUDA tensor on the communicator's device")
        out = torch.empty((self.n_ranks,) + tuple(t.shape), dtype=t.dtype, device=t.device)
        st = torch.cuda.current_stream(t.device).cuda_stream
        check(lib().clb_comm_all_gather(self._h, C.c_void_p(t.data_ptr()), C.c_void_p(out.data_ptr()),
                                        i64(t.numel() * t.element_size()), C.c_void_p(st)))
        return out

    def all_reduce_max_(self, t):
        """in-place element-wise maximum of a contiguous float32 CUDA tensor over the ranks"""
        import torch
        if not (t.is_cuda and t.is_contiguous() and t.dtype == torch.float32 and t.device.index == self.device):
            raise ValueError("all_reduce_max_ needs a contiguous float32 CUDA tensor on the communicator's device")
        st = torch.cuda.current_stream(t.device).cuda_stream
        check(lib().clb_comm_all_reduce_max_f32(self._h, C.c_void_p(t.data_ptr()), i64(t.numel()), C.c_void_p(st)))
        return t

    def sync_bound_consts(self, searcher):
        """one error bound on every shard (see sync_bound_consts below), over this communicator"""
        check(lib().clb_searcher_sync_bound_consts(searcher._h, self._h))     # the round trip inside the library
        return searcher.bound_consts


def all_gather_packed(packed, group=None):
    """ONE all-gather for a rank's whole result.  `packed`: the uint8 tensor of `packed_topk_bytes` bytes that
    DeviceSearch wrote its pids and scores into -> (world, nbytes) uint8 tensor on every rank."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    out = torch.empty(world * packed.numel(), dtype=torch.uint8, device=packed.device)   # concatenation along dim 0
    dist.all_gather_into_tensor(out, packed, group=group)
    return out.view(world, packed.numel())


def merge_packed(gathered, B: int, k: int, out_p=None, out_s=None):
    """(world, nbytes) gathered packed blocks -> (B, k) merged top-k via clb_merge_topk_packed_device on torch's
    current stream."""
    import torch
    world = gathered.shape[0]
    if not gathered.is_cuda:      # host merge (gloo tests): unpack the blocks and reuse the unpacked path
        gp = gathered[:, :B * k * 8].contiguous().view(torch.int64).view(world, B, k)
        gs = gathered[:, B * k * 8:B * k * 12].contiguous().view(torch.float32).view(world, B, k)
        return merge_gathered(gp, gs, k)
    if out_p is None:
        out_p = torch.empty((B, k), dtype=torch.int64, device=gathered.device)
        out_s = torch.empty((B, k), dtype=torch.float32, device=gathered.device)
    st = torch.cuda.current_stream(gathered.device).cuda_stream
    check(lib().clb_merge_topk_packed_device(gathered.device.index, C.c_void_p(gathered.data_ptr()), i64(k), i64(world),
                                             i64(B), C.c_void_p(out_p.data_ptr()), C.c_void_p(out_s.data_ptr()),
                                             C.c_void_p(st)))
    return out_p, out_s


def sync_bound_consts(searcher, group=None):
    """The two-phase sharded search cuts every shard at one global threshold tau - 2 eps, so eps must be the same
    (the largest) on every shard: all-reduce MAX of the three constants of the error bound, once, at load time.
    Without it a shard whose own embeddings give a smaller eps could skip a member of the global top-k."""
    import torch
    import torch.distributed as dist
    backend = dist.get_backend(group)
    dev = torch.device("cuda", searcher.device) if backend == "nccl" else torch.device("cpu")
    t = torch.from_numpy(searcher.bound_consts.copy()).to(dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    searcher.raise_bound_consts(t.cpu().numpy())
    return searcher.bound_consts


def share_bound_consts(searchers):
    """The same for several shards held by ONE process (tests, tools/shard_breakdown.py)."""
    m = np.max(np.stack([s.bound_consts for s in searchers]), axis=0)
    for s in searchers:
        s.raise_bound_consts(m)
    return m


def all_gather_scores(local_top, group=None):
    """(B, k) fp32 -> (world, B, k): the exchange between the two phases of the sharded search."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    B, k = local_top.shape
    out = torch.empty((world * B, k), dtype=local_top.dtype, device=local_top.device)
    dist.all_gather_into_tensor(out, local_top.contiguous(), group=group)
    return out.view(world, B, k)


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


def pack_topk(pids, scores):
    """(B, k) int64 pids + (B, k) fp32 scores -> the packed uint8 block of `packed_topk_bytes(k, B)` bytes."""
    import torch
    B, k = pids.shape
    buf = torch.zeros(packed_topk_bytes(k, B), dtype=torch.uint8, device=pids.device)
    buf[:B * k * 8].view(torch.int64).copy_(pids.reshape(-1))
    buf[B * k * 8:B * k * 12].view(torch.float32).copy_(scores.reshape(-1))
    return buf


def sharded_search(local_search, Q, k: int, group=None, packed: bool = False):
    """local_search(Q) -> (pids (B, k), scores (B, k)) torch tensors for this rank's shard, padded with
    (0, -inf).  Returns the merged (B, k) result, identical on every rank.  `packed`: move pids and scores in ONE
    all-gather of packed blocks (what bench.py does) instead of one collective each."""
    p, s = local_search(Q)
    if packed:
        return merge_packed(all_gather_packed(pack_topk(p, s), group), p.shape[0], k)
    gp, gs = all_gather_topk(p, s, group)
    return merge_gathered(gp, gs, k)


class DeviceSearch:
    """Device-resident batched search on one shard: queries and results stay in HBM (torch tensors),
    work is enqueued on torch's current stream."""

    def __init__(self, searcher, T: int, B: int, k: int, nprobe: int, slot: int = 0):
        import torch
        self.s, self.T, self.B, self.k, self.nprobe, self.slot = searcher, T, B, k, nprobe, slot
        self.dev = torch.device("cuda", searcher.device)
        # pids and scores live in one packed block so that a single all-gather can move both (all_gather_packed)
        self.packed = torch.empty(packed_topk_bytes(k, B), dtype=torch.uint8, device=self.dev)
        self.out_p = self.packed[:B * k * 8].view(torch.int64).view(B, k)
        self.out_s = self.packed[B * k * 8:B * k * 12].view(torch.float32).view(B, k)
        self.ncand = torch.zeros(B, dtype=torch.int64, device=self.dev)

    def _check_queries(self, Qdev):
        """The C ABI takes a bare pointer: a tensor with fewer than B x T x dim contiguous floats on this device would
        be read past its end by the kernels, so it is refused here."""
        import torch
        from ._lib import ArgumentError
        want = self.B * self.T * int(self.s.dim)
        if (not isinstance(Qdev, torch.Tensor) or Qdev.dtype != torch.float32 or Qdev.device != self.dev
                or not Qdev.is_contiguous() or Qdev.numel() != want):
            raise ArgumentError(f"queries must be a contiguous float32 tensor of {self.B} x {self.T} x {int(self.s.dim)} "
                                f"elements on {self.dev}, got {getattr(Qdev, 'dtype', type(Qdev))} "
                                f"{tuple(getattr(Qdev, 'shape', ()))} on {getattr(Qdev, 'device', '?')}")

    def phase1(self, Qdev):
        """Two-phase sharded search, first half (clb_search_shard_phase1): everything up to pass 1 on this shard.
        Returns the (B, k) tensor of the shard's k largest approximate scores per query, to be all-gathered."""
        import torch
        self._check_queries(Qdev)
        if not hasattr(self, "local_top"):
            self.local_top = torch.empty((self.B, self.k), dtype=torch.float32, device=self.dev)
        st = torch.cuda.current_stream(self.dev).cuda_stream
        check(lib().clb_search_shard_phase1_slot(self.s._h, C.c_int(self.slot), C.c_void_p(Qdev.data_ptr()), i64(self.T), i64(self.B),
                                            i64(self.nprobe), i64(self.k), C.c_void_p(self.local_top.data_ptr()),
                                            C.c_void_p(st)))
        return self.local_top

    def phase2(self, Qdev, all_top):
        """Second half (clb_search_shard_phase2).  all_top: (n_shards, B, k) gathered `local_top` blocks."""
        import torch
        from ._lib import ArgumentError
        self._check_queries(Qdev)
        if (all_top.dtype != torch.float32 or all_top.dim() != 3 or tuple(all_top.shape[1:]) != (self.B, self.k)
                or not all_top.is_contiguous() or all_top.device != self.dev):
            raise ArgumentError(f"all_top must be a contiguous float32 (n_shards, {self.B}, {self.k}) tensor on {self.dev}")
        st = torch.cuda.current_stream(self.dev).cuda_stream
        check(lib().clb_search_shard_phase2_slot(self.s._h, C.c_int(self.slot), C.c_void_p(Qdev.data_ptr()), i64(self.T), i64(self.B),
                                            i64(self.nprobe), i64(self.k), C.c_void_p(all_top.data_ptr()),
                                            i64(all_top.shape[0]), C.c_void_p(self.out_p.data_ptr()),
                                            C.c_void_p(self.out_s.data_ptr()), C.c_void_p(self.ncand.data_ptr()),
                                            C.c_void_p(st)))
        return self.out_p, self.out_s

    def capture(self, Qstatic):
        """A HIP graph of one search over the STATIC query buffer `Qstatic` (B, T, dim): the library enqueues nothing but
        kernels and memsets on the stream it is handed, so the nine launches of a search can be captured once and
        replayed -- write the next queries into `Qstatic` (e.g. let the encoder write there), `graph.replay()`, read
        `out_p` / `out_s`.  Measured (one query, 1 M passages; bench.py `p50_latency_graph_replay_ms`, tools/graph_probe.py):
        no faster than the stream launches (0.171-0.184 against 0.171-0.179 ms with a new query per replay) -- the host
        enqueues the eleven launches ahead of the device either way; on a 10 M-passage index, where a search is ~20
        launches, the replay is faster (0.259 against 0.318 ms).  The
        search runs once on a side stream first, so that the workspace is sized outside the capture."""
        import torch
        self._check_queries(Qstatic)
        cur = torch.cuda.current_stream(self.dev)
        side = torch.cuda.Stream(self.dev)
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            self(Qstatic)
        cur.wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            self(Qstatic)
        return graph

    def __call__(self, Qdev):
        """Qdev: torch float32 tensor holding B queries laid out (B, T, dim) contiguous == Julia (dim, T, B)."""
        import torch
        self._check_queries(Qdev)
        st = torch.cuda.current_stream(self.dev).cuda_stream
        check(lib().clb_search_batch_device_slot(self.s._h, C.c_int(self.slot), C.c_void_p(Qdev.data_ptr()), i64(self.T),
                                                 i64(self.B), i64(self.nprobe), i64(self.k),
                                                 C.c_void_p(self.out_p.data_ptr()), C.c_void_p(self.out_s.data_ptr()),
                                                 C.c_void_p(self.ncand.data_ptr()), C.c_void_p(st)))
        return self.out_p, self.out_s
