#!/usr/bin/env python3
"""bench.py -- queries/sec (+ p50 latency) of top-1000 search on a synthetic 1M-passage corpus
(BASELINE.json: dim 128, nbits 2, doclen ~80, query_maxlen 32, nprobe 2, k 1000) on N MI355X.

A step = one pass of the search hot path (centroid scoring -> candidates -> fused decompress+MaxSim
-> top-k, and for N > 1 the RCCL all-gather + merge of the per-shard top-k) over one batch of queries
that are already resident in HBM.  The passage collection is sharded over the N ranks (strong scaling:
the corpus is fixed).  Prints ONE JSON line on rank 0.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--docs D] [--batch B] [--mode {0,1}]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
F32_MFMA_PEAK_TF = 157.3     # MI355X_MICROARCH.md: fp32 matrix peak
BF16_MFMA_PEAK_TF = 2500.0   # MI355X_MICROARCH.md: dense bf16 matrix peak
BYTES_PER_EMB = 36.0         # 4-B code + 32-B packed residual
BYTES_PER_PID = 16.0
FLOP_PER_EMB = 8192.0        # 2 * 32 * 128


def launch_ranks(args):
    """`python bench.py --gpus N` without a launcher: start N fresh rank processes (one per GPU) BEFORE this process
    has touched HIP or imported torch, wait for them, and forward rank 0's single JSON line.  Never re-execs."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    # wait for all ranks; a rank that dies must not leave the others waiting in a collective forever
    import threading
    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    failed = None
    while any(p.poll() is None for p in procs):
        bad = [(r, p.returncode) for r, p in enumerate(procs) if p.poll() not in (None, 0)]
        if bad:
            failed = bad[0]
            print(f"[bench] rank {failed[0]} exited with code {failed[1]}: stopping the other ranks", file=sys.stderr)
            for p in procs:
                if p.poll() is None:
                    p.kill()
            break
        time.sleep(0.2)
    rcs = [p.wait() for p in procs]
    reader.join(timeout=10)
    out = b"".join(c for c in chunks if c)
    if failed is None:
        sys.stdout.write(out.decode())
        sys.stdout.flush()
    return failed[1] if failed else next((rc for rc in rcs if rc), 0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--docs", type=int, default=1_000_000)
    ap.add_argument("--batch", type=int, default=0,
                    help="queries per step; 0 = 32 per GPU, at most 256 (the centroid stage is replicated on every shard and the "
                         "selection kernels are one work-group per query: larger batches amortise both once the corpus is sharded)")
    ap.add_argument("--k", type=int, default=1000)
    ap.add_argument("--nprobe", type=int, default=2)
    ap.add_argument("--mode", type=int, default=-1, help="-1 library default, 0 exact, 1 two-pass")
    ap.add_argument("--cpu-seconds", type=float, default=20.0, help="budget of the cpu_baseline sample")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-latency", action="store_true", help="skip the one-query-at-a-time latency loop (profiling runs)")
    ap.add_argument("--force-gather", action="store_true", help="exercise the all-gather + merge path even with one rank (testing)")
    ap.add_argument("--uniform-codes", action="store_true",
                    help="passages draw their centroid codes uniformly (worst case for the candidate count) instead of topically")
    ap.add_argument("--no-overlap", action="store_true", help="one compute stream: batches strictly one after the other")
    ap.add_argument("--in-flight", type=int, default=2, help="batches in flight (compute streams / workspace slots), 1..4")
    ap.add_argument("--no-encoder", action="store_true",
                    help="skip the second measurement with the query encoder (bert-base geometry) in front of the search")
    ap.add_argument("--min-seconds", type=float, default=0.5,
                    help="the timed region is repeated (whole multiples of --steps) until it lasts at least this long")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args))

    # stdout carries exactly ONE line (the JSON): anything libraries print while we run (RCCL prints a version
    # banner on communicator creation) is sent to stderr
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    import torch
    import colbert_jl_amd as clb
    from colbert_jl_amd import synthetic
    from colbert_jl_amd.distributed import DeviceSearch, all_gather_packed, all_gather_scores, merge_packed

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 or args.force_gather:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        # COLBERT_BENCH_BACKEND / COLBERT_BENCH_DEVICE: test hooks to run several ranks on ONE GPU over gloo
        # (RCCL refuses two ranks on the same device); the measured configuration is always nccl, one GPU per rank
        backend = os.environ.get("COLBERT_BENCH_BACKEND", "nccl")
        if "COLBERT_BENCH_DEVICE" in os.environ:
            local_rank = int(os.environ["COLBERT_BENCH_DEVICE"])
        if backend == "nccl":
            dist.init_process_group(backend="nccl", rank=rank, world_size=world,
                                    device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=backend, rank=rank, world_size=world)
    if world != args.gpus:
        print(f"[bench] --gpus {args.gpus} but WORLD_SIZE={world}: refusing to report a number for a different job size",
              file=sys.stderr)
        sys.exit(2)
    ranks_seen = world
    if world > 1 or args.force_gather:
        ranks_seen = dist.get_world_size()
        if ranks_seen != args.gpus and not args.force_gather:
            print(f"[bench] the process group has {ranks_seen} ranks, --gpus {args.gpus}", file=sys.stderr)
            sys.exit(2)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    # ---- this rank's passage shard (generated directly, identical to the same passages of the full index)
    T, B, k = 32, (args.batch if args.batch > 0 else min(32 * world, 256)), args.k
    n_blocks = 8
    assert n_blocks % world == 0, "shards are aligned to the 8 generation blocks: use 1, 2, 4 or 8 GPUs"
    per = n_blocks // world
    t0 = time.time()
    K = synthetic.num_partitions_for(args.docs, 80.0)
    shard = synthetic.make_index(seed=2024, n_docs=args.docs, K=K, n_blocks=n_blocks,
                                 blocks=range(rank * per, (rank + 1) * per), topical=not args.uniform_codes)
    t_gen = time.time() - t0
    t0 = time.time()
    s = clb.Searcher(index=shard, device=local_rank, pid_offset=int(shard["pid_offset"]))
    if args.mode >= 0:
        s.set_mode(args.mode)
    if world > 1:
        from colbert_jl_amd.distributed import sync_bound_consts
        sync_bound_consts(s)            # one error bound on every shard (the threshold of the two-phase search is global)
    t_load = time.time() - t0
    n_queries = max(B * 8, 256)
    Q = synthetic.make_topic_queries(shard["centroids"], seed=77, n_queries=n_queries, T=T)   # (dim, T, nq)
    Qdev = torch.from_numpy(np.ascontiguousarray(Q.transpose(2, 1, 0))).to(dev)               # (nq, T, dim)
    gather = world > 1 or args.force_gather
    # Two result buffers alternate so that the exchange of batch i (one RCCL all-gather of the packed per-shard
    # top-k + the merge kernel, on a side stream) overlaps the search of batch i+1 on the main stream.
    # Two batches in flight: batch i runs on compute stream i % NF with its own workspace slot and result buffers, so
    # the latency-bound kernels of one batch (selection, top-k: one work-group per query) overlap the scoring kernels
    # of the other.  --no-overlap puts every batch on one stream.
    NF = max(1, min(4, args.in_flight))
    runs = [DeviceSearch(s, T, B, k, args.nprobe, slot=i) for i in range(NF)]
    run = runs[0]
    compute = [torch.cuda.Stream(device=dev) for _ in range(NF)]
    merged = [(torch.empty((B, k), dtype=torch.int64, device=dev), torch.empty((B, k), dtype=torch.float32, device=dev))
              for _ in range(NF)]
    comm = torch.cuda.Stream(device=dev) if gather else None
    free_ev = [None] * NF          # buffer set i may be overwritten once its exchange has finished
    # With several shards the search runs in two phases around a second, small all-gather (every shard's k largest
    # approximate scores): all shards then cut at the GLOBAL k-th score and re-score ~k/N passages each instead of
    # ~k.  COLBERT_BENCH_TWO_PHASE=0/1 overrides.
    two_phase = gather and s.mode == 1 and (world >= 2 if "COLBERT_BENCH_TWO_PHASE" not in os.environ
                                            else os.environ["COLBERT_BENCH_TWO_PHASE"] == "1")
    import torch.distributed as _dist
    gath = [torch.empty((max(world, 1) * B, k), dtype=torch.float32, device=dev) for _ in range(NF)] if two_phase else None

    def search_shard(r, Qb, i):
        """This rank's part of one batch on the main stream (results in r.packed)."""
        if not two_phase:
            r(Qb)
            return
        main = torch.cuda.current_stream(dev)
        lt = r.phase1(Qb)
        e1 = torch.cuda.Event()
        e1.record(main)
        with torch.cuda.stream(comm):
            comm.wait_event(e1)
            _dist.all_gather_into_tensor(gath[i % NF], lt)
            e2 = torch.cuda.Event()
            e2.record(comm)
        main.wait_event(e2)
        r.phase2(Qb, gath[i % NF].view(world, B, k))

    overlap = [not args.no_overlap]

    def step(i):
        with torch.cuda.stream(compute[i % NF] if overlap[0] else compute[0]):
            return step_on_current_stream(i)

    def step_on_current_stream(i):
        off = (i * B) % (n_queries - B + 1)
        r = runs[i % NF]
        if not gather:
            return r(Qdev[off:off + B])
        main = torch.cuda.current_stream(dev)
        if free_ev[i % NF] is not None:
            main.wait_event(free_ev[i % NF])
        search_shard(r, Qdev[off:off + B], i)
        done = torch.cuda.Event()
        done.record(main)
        with torch.cuda.stream(comm):
            comm.wait_event(done)
            g = all_gather_packed(r.packed)
            out = merge_packed(g, B, k, out_p=merged[i % NF][0], out_s=merged[i % NF][1])
            ev = torch.cuda.Event()
            ev.record(comm)
            free_ev[i % NF] = ev
        return out

    def barrier():
        torch.cuda.synchronize()
        if gather:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(n_steps, first):
        """n_steps steps between two barrier + synchronize pairs; MAX over ranks of the wall time."""
        barrier()
        t0 = time.perf_counter()
        for i in range(n_steps):
            step(first + i)
        barrier()
        dt = time.perf_counter() - t0
        if world > 1:
            tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dt = float(tmax.item())
        return dt

    # (0) pre-conditioning + the sustained figure: the same loop repeated in whole multiples of --steps until the region
    #     lasts --min-seconds (a 20-step region is ~25 ms: too short to trust on its own, and a GPU that has just
    #     come out of index generation idles at a low clock); every rank uses the same count
    for i in range(args.warmup):
        step(i)
    probe = timed(args.steps, args.warmup)
    reps = max(1, int(np.ceil(args.min_seconds / max(probe, 1e-6))))
    if world > 1:
        r_t = torch.tensor([reps], dtype=torch.int64, device=dev)
        dist.all_reduce(r_t, op=dist.ReduceOp.MAX)
        reps = int(r_t.item())
    sustained_steps = reps * args.steps
    sustained_s = timed(sustained_steps, args.warmup + args.steps)
    # (1) the contract's measurement on the warm device: --warmup untimed steps, then exactly --steps steps between
    #     barrier + synchronize pairs, per-kernel event timing OFF
    for i in range(args.warmup):
        step(i)
    elapsed = timed(args.steps, args.warmup)
    qps = B * args.steps / elapsed
    # (3) a separate pass with HIP events around every kernel (on the stream they are launched on) for the roofline
    #     -- on ONE stream, so that a kernel's time is its own (with two batches in flight kernels share the chip)
    was = overlap[0]
    overlap[0] = False
    serial_s = timed(args.steps, args.warmup)
    s.profile_enable(True)
    prof_steps = args.steps
    timed(prof_steps, args.warmup)
    prof = s.profile_read()
    s.profile_enable(False)
    overlap[0] = was

    # ---- batches in flight do not disturb each other: one batch computed alone == the same batch computed in the middle
    # of NF batches in flight on NF streams (every rank checks its own results)
    def result_of(i):
        out = step(i)
        st = comm if gather else (compute[i % NF] if overlap[0] else compute[0])
        with torch.cuda.stream(st):
            return out[0].clone(), out[1].clone()
    probe_i = args.warmup + 2 * NF + 1
    was = overlap[0]
    overlap[0] = False
    barrier()
    alone = result_of(probe_i)
    barrier()
    overlap[0] = was
    for i in range(probe_i - NF, probe_i):
        step(i)
    busy = result_of(probe_i)
    for i in range(probe_i + 1, probe_i + 1 + NF):
        step(i)
    barrier()
    in_flight_ok = bool(torch.equal(alone[0], busy[0]) and torch.equal(alone[1], busy[1]))
    if world > 1:
        okt = torch.tensor([1 if in_flight_ok else 0], dtype=torch.int64, device=dev)
        dist.all_reduce(okt, op=dist.ReduceOp.MIN)
        in_flight_ok = bool(okt.item())

    # ---- the metric as the reference's search(::String) defines it (src/searching.jl:93-128): encode_queries first.
    # No checkpoint exists in the build image, so the encoder has bert-base-uncased GEOMETRY with random weights and
    # runs on synthetic token ids; random weights give meaningless embeddings that would change the candidate
    # statistics, so the search half of the step still consumes the synthetic queries of the headline line: the
    # step costs exactly "encode B queries + search B queries", back to back on one stream.
    e2e = None
    if not args.no_encoder:
        from colbert_jl_amd.encoder import BERT_BASE, random_weights
        enc = clb.BertEncoder(random_weights(BERT_BASE, 128, seed=5), dict(BERT_BASE), dim=128, device=local_rank)
        rng = np.random.default_rng(6)
        d_ids = torch.from_numpy(rng.integers(1, BERT_BASE["vocab_size"] + 1, size=(n_queries, T)).astype(np.int32)).to(dev)
        d_mask = torch.ones((n_queries, T), dtype=torch.uint8, device=dev)
        d_skip = torch.tensor([1], dtype=torch.int64, device=dev)
        q_enc = torch.empty((B, T, 128), dtype=torch.float32, device=dev)

        q_encs = [q_enc] + [torch.empty_like(q_enc) for _ in range(NF - 1)]
        enc_done = [None]          # the encoder has ONE activation workspace: its calls are chained by an event

        def step_e2e(i):
            off = (i * B) % (n_queries - B + 1)
            st = compute[i % NF] if overlap[0] else compute[0]
            with torch.cuda.stream(st):
                if enc_done[0] is not None:
                    st.wait_event(enc_done[0])
                enc.query_embeddings_device(d_ids[off:off + B], d_mask[off:off + B], d_skip, q_encs[i % NF])
                ev = torch.cuda.Event()
                ev.record(st)
                enc_done[0] = ev
                return step_on_current_stream(i)

        for i in range(3):
            step_e2e(i)
        barrier()
        t0 = time.perf_counter()
        for i in range(args.steps):
            step_e2e(args.warmup + i)
        barrier()
        dt = time.perf_counter() - t0
        t0 = time.perf_counter()
        for i in range(args.steps):
            off = (i * B) % (n_queries - B + 1)
            enc.query_embeddings_device(d_ids[off:off + B], d_mask[off:off + B], d_skip, q_enc)
        barrier()
        dt_enc = time.perf_counter() - t0
        if world > 1:
            tm = torch.tensor([dt, dt_enc], dtype=torch.float64, device=dev)
            dist.all_reduce(tm, op=dist.ReduceOp.MAX)
            dt, dt_enc = float(tm[0].item()), float(tm[1].item())
        e2e = {"value": round(B * args.steps / dt, 2), "unit": "queries/s", "ms_per_step": round(dt / args.steps * 1e3, 4),
               "encoder_ms_per_step": round(dt_enc / args.steps * 1e3, 4),
               "dtype": "f32" if enc.gemm == "f32" else f"f32 ({enc.gemm}: every fp32 product of the Linear layers as exact bf16 MFMA products of split operands, fp32 accumulation)",
               "encoder": "bert-base-uncased geometry (12 x 768, 12 heads, FFN 3072) + Dense 768->128, random weights, "
                          "synthetic token ids; every rank encodes the whole batch",
               "note": "encode_queries + search per step; the search consumes the synthetic queries of the headline line"}
        enc.close()

    # ---- work counters of one batch (for the roofline) and p50 latency, outside the timed region
    s.profile_enable(True, counters=True)
    step(args.warmup)
    torch.cuda.synchronize()
    stats = s.last_batch_stats()
    s.profile_read()
    s.profile_enable(False)
    lat = []
    one = DeviceSearch(s, T, 1, k, args.nprobe)
    for i in range(0 if args.no_latency else 40):
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        if two_phase:
            one.phase2(Qdev[i:i + 1], all_gather_scores(one.phase1(Qdev[i:i + 1])))
        else:
            one(Qdev[i:i + 1])
        if gather:
            merge_packed(all_gather_packed(one.packed), 1, k)
        torch.cuda.synchronize()
        lat.append(time.perf_counter() - t1)
    p50_ms = float(np.median(lat[5:]) * 1e3) if lat else None

    # ---- roofline of the dominant kernel (per launch = one batch on this rank's shard)
    dom = max(prof.items(), key=lambda kv: kv[1]["ms"])[0] if prof else None
    roof = None
    if dom:
        def roof_of(kname):
            ms_launch = prof[kname]["ms"] / max(prof[kname]["launches"], 1)
            embs = stats["cand_embs"]; docs = stats["cand_docs"]
            if kname == "score_exact" and s.mode == 1:
                embs, docs = stats["rescored_embs"], stats["rescored_docs"]
            if kname == "score_exact":
                ach = FLOP_PER_EMB * embs / (ms_launch * 1e-3) / 1e12
                r = {"kernel": kname, "bound": "mfma", "achieved": round(ach, 2), "peak": F32_MFMA_PEAK_TF,
                     "unit": "TFLOP/s", "frac": round(ach / F32_MFMA_PEAK_TF, 4)}
            elif kname == "centroid_scores":
                # S1 runs as three bf16 MFMA products per fp32 product (bf16x3 split): count the bf16 flops it
                # really issues against the dense bf16 peak
                ach = 3 * 2.0 * 128 * T * K * B / (ms_launch * 1e-3) / 1e12
                r = {"kernel": kname, "bound": "mfma", "achieved": round(ach, 2), "peak": BF16_MFMA_PEAK_TF,
                     "unit": "TFLOP/s (bf16, 3 products per fp32 product)", "frac": round(ach / BF16_MFMA_PEAK_TF, 4)}
                embs, docs = K * B, 0
            else:
                alg_bytes = BYTES_PER_EMB * embs + BYTES_PER_PID * docs
                ach = alg_bytes / (ms_launch * 1e-3) / 1e9
                r = {"kernel": kname, "bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(ach / HBM_PEAK_GBS, 4)}
            r["ms_per_launch"] = round(ms_launch, 4)
            r["units_per_launch"] = {"embeddings": int(embs), "passages": int(docs)}
            return r
        roof = roof_of(dom)
        # HBM bytes per launch: PMC counters cannot be read from inside this process; they come from the committed
        # rocprofv3 --pmc passes of this same command (tools/pmc_summary.py), which record the hash of the kernel
        # sources they were measured on.  A summary taken on different sources is stale: traffic stays null.
        roof["traffic"] = None
        pmc_file = os.path.join(ROOT, "profiles", "pmc_summary.json")
        # ... and on THIS workload (tools/gpu_profile.sh profiles the default one): bytes per launch do not transfer
        default_workload = (args.docs == 1_000_000 and not args.uniform_codes and B == 32 and k == 1000
                            and args.nprobe == 2 and args.mode < 0)
        if world == 1 and not default_workload:
            roof["traffic_source"] = "null: profiles/pmc_summary.json was measured on the default workload, not this one"
        elif world == 1 and os.path.exists(pmc_file):
            from tools.pmc_summary import csrc_hash
            pmc_all = json.load(open(pmc_file))
            pmc = pmc_all.get("kernels", {}).get(dom, {})
            if pmc_all.get("csrc_sha256") == csrc_hash() and "hbm_read_bytes" in pmc:
                raw = pmc["hbm_read_bytes_uncorrected"]
                if dom == "score_approx":
                    # FETCH_SIZE counts this kernel's contiguous streams (residual 32 B + one 4-B code|inv_norm word per
                    # embedding) at half their bytes and its 64-B score-row gathers in full (fetch_calib.hip)
                    stream = 36.0 * stats["cand_embs"]
                    roof["traffic"] = int(raw + 0.5 * stream + pmc.get("hbm_write_bytes", 0))
                    roof["traffic_split"] = {"stream_bytes": int(stream), "gather_miss_bytes": int(raw - 0.5 * stream)}
                else:
                    roof["traffic"] = pmc["hbm_read_bytes"] + pmc.get("hbm_write_bytes", 0)
                roof["traffic_source"] = ("profiles/pmc_summary.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of "
                                          "this command; " + pmc_all.get("correction", ""))
            else:
                roof["traffic_source"] = "null: profiles/pmc_summary.json was measured on different kernel sources"
        roof["all_kernels_ms_per_step"] = {kname: round(v["ms"] / max(prof_steps, 1), 4) for kname, v in prof.items()}
        roof["other_kernels"] = [roof_of(kn) for kn in ("score_approx", "score_exact", "centroid_scores")
                                 if kn in prof and kn != dom and prof[kn]["launches"]]

    # ---- CPU baseline: the oracle (a port of the reference algorithm) on the host cores, rank 0, N = 1
    cpu = None
    if world == 1 and not args.no_cpu:
        from oracle import oracle as orc
        orc.build()
        emb2pid = orc.build_emb2pid(shard["doclens"])
        idx = dict(shard, emb2pid=emb2pid)
        nq_cpu, t_cpu, ok = 0, 0.0, True
        search_shard(run, Qdev[0:B], 0); p, sc = run.out_p, run.out_s; torch.cuda.synchronize()
        gp_host = p.cpu().numpy(); gs_host = sc.cpu().numpy()
        while nq_cpu < B and (t_cpu < args.cpu_seconds or nq_cpu == 0):
            t1 = time.perf_counter()
            rp, rs, _ = orc.search(idx, Q[:, :, nq_cpu], args.nprobe, k)
            t_cpu += time.perf_counter() - t1
            ok = ok and bool(np.array_equal(rp, gp_host[nq_cpu])) and bool(np.max(np.abs(rs - gs_host[nq_cpu])) <= 1e-4)
            nq_cpu += 1
        cpu = {"value": round(nq_cpu / t_cpu, 4), "unit": "queries/s", "cores": orc.num_threads(), "kind": "port",
               "sample": f"{nq_cpu} queries of the same workload, one at a time, OpenMP over the host cores",
               "gpu_matches_cpu_top_k": ok}

    if rank == 0:
        out = {"metric": "queries/sec, top-1000 on 1M-passage corpus", "value": round(qps, 2), "unit": "queries/s",
               "n_gpus": world, "ranks_seen": ranks_seen, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "strong",
               "vs_baseline": None, "dtype": "f32", "data": "synthetic",
               "config": {"workload": f"synthetic {args.docs} passages (dim 128, nbits 2, doclen~80, K={K}"
                                      f"{', uniform codes' if args.uniform_codes else ''}), "
                                      f"top-{k}, nprobe {args.nprobe}, query_maxlen {T}, batch {B} queries/step, "
                                      f"passages sharded over {world} GPU(s)",
                          "search_mode": ("two-pass (bf16 MFMA prefilter + exact fp32 re-score)" if s.mode == 1 else "exact fp32 single pass")
                                         + (", global threshold exchange between the passes" if two_phase else "")},
               "batches_in_flight": NF if overlap[0] else 1, "in_flight_matches_serial": in_flight_ok,
               "one_batch_at_a_time": {"value": round(B * args.steps / serial_s, 2), "ms_per_step": round(serial_s / args.steps * 1e3, 4)},
               "sustained": {"steps": sustained_steps, "seconds": round(sustained_s, 4),
                             "value": round(B * sustained_steps / sustained_s, 2),
                             "note": "the same loop repeated until the timed region lasts --min-seconds"},
               "end_to_end_with_query_encoder": e2e,
               "p50_latency_ms": None if p50_ms is None else round(p50_ms, 4), "roofline": roof, "cpu_baseline": cpu,
               "setup_seconds": {"generate": round(t_gen, 1), "upload_and_build": round(t_load, 1)},
               "hbm_bytes": s.device_bytes}
        sys.stdout.flush()
        os.dup2(real_stdout, 1)
        print(json.dumps(out), flush=True)
        os.dup2(2, 1)
    s.close()
    if gather:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
