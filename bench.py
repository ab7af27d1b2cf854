#!/usr/bin/env python3
"""bench.py -- queries/sec (+ p50 latency) of top-1000 search on a synthetic 1M-passage corpus
(BASELINE.json: dim 128, nbits 2, doclen ~80, query_maxlen 32, nprobe 2, k 1000) on N MI355X.

A step = one pass of the search hot path (centroid scoring -> candidates -> fused decompress+MaxSim
-> top-k, and for N > 1 the RCCL all-gather + merge of the per-shard top-k) over one batch of queries
that are already resident in HBM.  The passage collection is sharded over the N ranks (strong scaling:
the corpus is fixed).  Prints ONE JSON line on rank 0.

Besides the headline the line carries (N = 1): `worst_case_uniform_codes` (the same corpus size with uniformly drawn
centroid codes: no id-adjacent codes, the largest candidate sets), `built_index` (BASELINE config 2: 100k passages
of mixture embeddings -> this repo's own k-means / codec / compress / IVF build -> search, oracle-checked, with an
`index_build` record) and `batch_sweep` (N = 1 at the batch sizes the N-GPU runs use); for N > 1: `fixed_batch_32`
(the same 32-query batch at every N) and `single_exchange` (one all-gather per batch instead of two).

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
GUIDE_COPY_GBS = 6290.0      # ... and the copy rate the guide quotes for this part; the line carries the rate MEASURED in the run
MEASURED_READ = {}           # device -> GB/s of a read-only sweep over 1 GB (measured_read_rate), filled beside MEASURED_COPY
MEASURED_COPY = {}           # device -> GB/s of a 1-GB device-to-device copy kernel, measured once per process (measured_copy_rate)


def measured_copy_rate(clb, device):
    """GB/s of clb_measure_copy_rate on this device: five 1-GB copies between two HIP events, once per run (~2 ms of copies)."""
    if device not in MEASURED_COPY:
        from colbert_jl_amd._lib import measure_copy_rate, measure_read_rate
        MEASURED_COPY[device] = round(measure_copy_rate(device, 1 << 30, 5), 1)
        MEASURED_READ[device] = round(measure_read_rate(device, 1 << 30, 5), 1)
    return MEASURED_COPY[device]
F32_MFMA_PEAK_TF = 157.3     # MI355X_MICROARCH.md: fp32 matrix peak
BF16_MFMA_PEAK_TF = 2500.0   # MI355X_MICROARCH.md: dense bf16 matrix peak


def nearest_products():
    """Products per fp32 product of the index build's nearest-centroid group lists: ONE fp16 product since round 5
    (nearest_top_f16_dma_kernel), three bf16 ones when COLBERT_NEAREST_PRODUCTS=3 asks for the earlier kernel.  Both run at
    the same dense 16-bit MFMA peak."""
    n = 3 if os.environ.get("COLBERT_NEAREST_PRODUCTS") == "3" else 1
    return n, ("bf16, 3 products per fp32 product" if n == 3 else "fp16, 1 product per fp32 product")

BYTES_PER_EMB = 36.0         # 4-B code + 32-B packed residual
BYTES_PER_PID = 16.0
FLOP_PER_EMB = 8192.0        # 2 * 32 * 128


def launch_ranks(args):
    """`python bench.py --gpus N` without a launcher: start N fresh rank processes (one per GPU) BEFORE this process
    has touched HIP or imported torch, wait for them, and forward rank 0's single JSON line.  Never re-execs.  The
    ranks meet through a file store (no port is picked here, so there is nothing to race for)."""
    import subprocess
    import tempfile
    import threading
    store = tempfile.NamedTemporaryFile(prefix="colbert_bench_store_", delete=False)
    store.close()
    os.unlink(store.name)
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus),
                   COLBERT_BENCH_INIT_FILE=store.name,
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    # wait for all ranks; a rank that dies must not leave the others waiting in a collective forever
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
    try:
        os.unlink(store.name)
    except OSError:
        pass
    out = b"".join(c for c in chunks if c)
    if failed is None:
        sys.stdout.write(out.decode())
        sys.stdout.flush()
    return failed[1] if failed else next((rc for rc in rcs if rc), 0)


def roofline_of(kname, prof, stats, mode, T, K, B, copy_gbs=None):
    """Roofline record of one search kernel: ALGORITHMIC bytes / flops of a launch over its HIP-event time.
    copy_gbs: the copy rate measured in this run on this device (measured_copy_rate)."""
    ms_launch = prof[kname]["ms"] / max(prof[kname]["launches"], 1)
    embs, docs = stats["cand_embs"], stats["cand_docs"]
    if kname == "score_exact" and mode == 1:
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
        if copy_gbs:
            r["measured_copy_rate"] = copy_gbs
            r["measured_copy_rate_how"] = "clb_measure_copy_rate in this run: 5 x 1 GB device-to-device between HIP events, best of three 16-B-per-lane kernel forms and hipMemcpyAsync"
            r["frac_of_measured_copy_rate"] = round(ach / copy_gbs, 4)
            rd = next(iter(MEASURED_READ.values()), None) if len(MEASURED_READ) == 1 else None
            if rd:
                r["measured_read_rate"] = rd
        r["guide_copy_rate"] = GUIDE_COPY_GBS
    r["ms_per_launch"] = round(ms_launch, 4)
    r["units_per_launch"] = {"embeddings": int(embs), "passages": int(docs)}
    return r


def traffic_from_pmc(pmc_file, dom, stats, roof):
    """roofline.traffic from a committed rocprofv3 --pmc summary (tools/gpu_profile.sh -> tools/pmc_summary.py) of THIS
    workload: counters cannot be read in-process.  The summary records the sha256 of the kernel sources it was measured
    on; a summary taken on different sources leaves traffic null."""
    if not os.path.exists(pmc_file):
        roof["traffic_source"] = f"null: no {os.path.relpath(pmc_file, ROOT)}"
        return
    from tools.pmc_summary import csrc_hash
    pmc_all = json.load(open(pmc_file))
    pmc = pmc_all.get("kernels", {}).get(dom, {})
    if pmc_all.get("csrc_sha256") != csrc_hash() or "hbm_read_bytes" not in pmc:
        roof["traffic_source"] = f"null: {os.path.relpath(pmc_file, ROOT)} was measured on different kernel sources"
        return
    raw = pmc["hbm_read_bytes_uncorrected"]
    if dom == "score_approx":
        # FETCH_SIZE counts this kernel's contiguous streams (residual 32 B + one 4-B code|inv_norm word per
        # embedding) at half their bytes and its 64-B score-row gathers in full (fetch_calib.hip)
        stream = 36.0 * stats["cand_embs"]
        roof["traffic"] = int(raw + 0.5 * stream + pmc.get("hbm_write_bytes", 0))
        roof["traffic_split"] = {"stream_bytes": int(stream), "gather_miss_bytes": int(raw - 0.5 * stream)}
    else:
        roof["traffic"] = pmc["hbm_read_bytes"] + pmc.get("hbm_write_bytes", 0)
    roof["traffic_source"] = (f"{os.path.relpath(pmc_file, ROOT)}: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this "
                              "workload; " + pmc_all.get("correction", ""))
    if roof.get("traffic") and roof.get("measured_read_rate"):
        # what the kernel REALLY moves per second (PMC bytes per launch over the HIP-event time of this run) against the read rate
        # measured in this run: how close the pass is to the memory system's own ceiling, wasted re-reads included
        tr = roof["traffic"] / (roof["ms_per_launch"] * 1e-3) / 1e9
        roof["traffic_rate_GBps"] = round(tr, 1)
        roof["traffic_frac_of_measured_read_rate"] = round(tr / roof["measured_read_rate"], 4)


def measure_sub(torch, clb, s, index, Q, B, k, nprobe, steps, min_seconds, cpu_queries, dev, in_flight=2, T=32, pmc_key=None):
    """A short single-GPU measurement of one workload (the sub-records of the line): sustained queries/s with
    `in_flight` batches in flight, the per-kernel HIP-event times of a one-batch-at-a-time pass with the roofline of
    pass 1, candidates per query, and `cpu_queries` queries checked against the CPU oracle (pids identical, scores
    within 1e-4)."""
    from colbert_jl_amd.distributed import DeviceSearch
    K = int(np.asarray(index["centroids"]).shape[1])
    n_queries = Q.shape[2]
    Qdev = torch.from_numpy(np.ascontiguousarray(Q.transpose(2, 1, 0))).to(dev)               # (nq, T, dim)
    NF = max(1, min(4, in_flight))
    runs = [DeviceSearch(s, T, B, k, nprobe, slot=i) for i in range(NF)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(NF)]

    def step(i, overlap=True):
        off = (i * B) % (n_queries - B + 1)
        with torch.cuda.stream(streams[i % NF] if overlap else streams[0]):
            return runs[i % NF](Qdev[off:off + B])

    def timed(n, overlap=True):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(n):
            step(i, overlap)
        torch.cuda.synchronize()
        return time.perf_counter() - t0

    for i in range(3):
        step(i)
    probe = timed(steps)
    reps = max(1, int(np.ceil(min_seconds / max(probe, 1e-6))))
    n = reps * steps
    dt = timed(n)
    serial = timed(steps, overlap=False)
    s.profile_enable(True)
    timed(steps, overlap=False)
    prof = s.profile_read()
    s.profile_enable(True, counters=True)
    step(0, overlap=False)
    torch.cuda.synchronize()
    stats = s.last_batch_stats()
    s.profile_read()
    s.profile_enable(False)
    rec = {"value": round(B * n / dt, 2), "unit": "queries/s", "ms_per_step": round(dt / n * 1e3, 4), "steps": n,
           "seconds": round(dt, 4), "batch": B, "batches_in_flight": NF,
           "one_batch_at_a_time": {"value": round(B * steps / serial, 2), "ms_per_step": round(serial / steps * 1e3, 4)},
           "candidates_per_query": {"passages": round(stats["cand_docs"] / B, 1), "embeddings": round(stats["cand_embs"] / B, 1),
                                    "rescored_passages": round(stats["rescored_docs"] / B, 1)}}
    if prof:
        dom = "score_approx" if "score_approx" in prof and prof["score_approx"]["launches"] else max(prof.items(), key=lambda kv: kv[1]["ms"])[0]
        roof = roofline_of(dom, prof, stats, s.mode, T, K, B, copy_gbs=measured_copy_rate(clb, dev.index or 0))
        roof["traffic"] = None
        if pmc_key:      # profiles/pmc_summary_<workload>.json: the PMC passes of this sub-record's workload (tools/gpu_profile.sh)
            traffic_from_pmc(os.path.join(ROOT, "profiles", f"pmc_summary_{pmc_key}.json"), dom, stats, roof)
        else:
            roof["traffic_source"] = "null: no PMC summary is kept for this workload"
        roof["all_kernels_ms_per_step"] = {kn: round(v["ms"] / max(steps, 1), 4) for kn, v in prof.items()}
        rec["roofline"] = roof
    if cpu_queries > 0:
        from oracle import oracle as orc
        orc.build()
        idx = dict(index, emb2pid=orc.build_emb2pid(index["doclens"]))
        r = runs[0]
        with torch.cuda.stream(streams[0]):
            r(Qdev[0:B])
        torch.cuda.synchronize()
        gp, gs = r.out_p.cpu().numpy(), r.out_s.cpu().numpy()
        ok, t_cpu = True, 0.0
        nq = min(cpu_queries, B)
        for j in range(nq):
            t1 = time.perf_counter()
            rp, rs, _ = orc.search(idx, Q[:, :, j], nprobe, k)
            t_cpu += time.perf_counter() - t1
            ok = ok and bool(np.array_equal(rp, gp[j])) and bool(np.max(np.abs(rs - gs[j])) <= 1e-4)
        rec["gpu_matches_cpu_top_k"] = ok
        rec["cpu_baseline"] = {"value": round(nq / t_cpu, 4), "unit": "queries/s", "cores": orc.num_threads(), "kind": "port",
                               "sample": f"{nq} queries of this workload, one at a time, OpenMP over the host cores"}
    return rec


def build_config2_index(clb, docs, kmeans_iters, device):
    """BASELINE config 2: `docs` passages of mixture embeddings through THIS repo's index build -- sample, k-means,
    codec statistics, compress, IVF (src/indexing.jl:63-147, collection_indexer.jl:219-237,349-353) -- timed per stage.
    The stand-alone entry points take host buffers, so their PCIe copies are inside the stage times."""
    from colbert_jl_amd import codec, synthetic
    t0 = time.time()
    embs, doclens = synthetic.make_embeddings(seed=61, n_docs=docs)
    t_gen = time.time() - t0
    n_emb = embs.shape[1]
    rng = np.random.default_rng(62)
    t0 = time.time()
    n_s = codec.num_sampled_pids(docs)
    off = np.concatenate([[0], np.cumsum(doclens)])
    pids = np.unique(rng.integers(0, docs, size=n_s))
    cols = np.concatenate([np.arange(off[p], off[p + 1]) for p in pids])
    sample = np.asfortranarray(embs[:, rng.permutation(cols)])
    h = codec.heldout_size(sample.shape[1])
    sample, held = sample[:, :-h], sample[:, -h:]
    plan = codec.setup(docs, float(doclens[pids].mean()), sample.shape[1], 25000, 1)
    K = plan["num_partitions"]
    init = sample[:, rng.permutation(sample.shape[1])[:K]]
    t_sample = time.time() - t0
    rec = {"passages": int(docs), "embeddings": int(n_emb), "sample_points": int(sample.shape[1]), "K": int(K),
           "generate_input_s": round(t_gen, 1), "sample_and_split_s": round(t_sample, 2)}
    t0 = time.time()
    cent, _, it = codec.kmeans(sample, init, max_iters=kmeans_iters, device=device)
    dt = time.time() - t0
    it = max(int(it), 1)
    rec["kmeans_s"] = round(dt, 3)
    rec["kmeans_iters"] = it
    rec["kmeans_s_per_iter"] = round(dt / it, 3)
    # assignment = the MFMA flops the group lists really execute, against the dense 16-bit peak
    n_prod, what = nearest_products()
    bf16_tf = n_prod * 2.0 * 128 * sample.shape[1] * K * it / dt / 1e12
    rec["kmeans_roofline"] = {"bound": "mfma", "achieved": round(bf16_tf, 1), "peak": BF16_MFMA_PEAK_TF,
                              "unit": f"TFLOP/s ({what}; host copies and the centroid update inside the time)",
                              "frac": round(bf16_tf / BF16_MFMA_PEAK_TF, 4)}
    t0 = time.time()
    cut, w, avg, _ = codec.compute_avg_residuals(2, cent, held, device=device)
    rec["codec_stats_s"] = round(time.time() - t0, 3)
    t0 = time.time()
    chunk = 2_000_000
    parts = [codec.compress(cent, cut, 128, 2, embs[:, i:i + chunk], device=device) for i in range(0, n_emb, chunk)]
    dt = time.time() - t0
    codes = np.concatenate([p[0] for p in parts])
    residuals = np.asfortranarray(np.concatenate([p[1] for p in parts], axis=1))
    rec["compress_s"] = round(dt, 3)
    rec["compress_Membeddings_per_s"] = round(n_emb / dt / 1e6, 2)
    t0 = time.time()
    ivf, lens = codec.build_ivf(codes, K, device=device)
    rec["build_ivf_s"] = round(time.time() - t0, 3)
    assert int(lens.sum()) == n_emb
    rec["total_build_s"] = round(rec["sample_and_split_s"] + rec["kmeans_s"] + rec["codec_stats_s"] + rec["compress_s"] + rec["build_ivf_s"], 2)
    index = {"dim": 128, "nbits": 2, "centroids": cent, "bucket_weights": w, "bucket_cutoffs": cut, "doclens": doclens,
             "codes": codes, "residuals": residuals, "ivf": ivf, "ivf_lengths": lens, "pid_offset": 0}
    return index, rec


def build_device_index(torch, docs, kmeans_iters, dev):
    """`docs` passages of mixture embeddings through this repo's index build with every large array resident in HBM
    (indexer.index_device: the fp32 embeddings of 1 M passages are 41 GB and exist only chunk by chunk, generated on the
    device): sample -> k-means -> codec statistics -> per chunk compress -> IVF, seconds per stage."""
    from colbert_jl_amd import indexer, synthetic
    src = synthetic.DeviceMixtureSource(seed=61, n_docs=docs, device=dev)
    index, rec = indexer.index_device(src, nbits=2, kmeans_niters=kmeans_iters, seed=62)
    for stage, n_pts, secs in (("kmeans", rec["sample_points"] * rec["kmeans_iters"], rec["kmeans_s"]),
                               ("compress", rec["embeddings"], rec["compress_s"])):
        n_prod, what = nearest_products()
        tf = n_prod * 2.0 * 128 * n_pts * rec["K"] / max(secs, 1e-9) / 1e12
        rec[stage + "_roofline"] = {"bound": "mfma", "achieved": round(tf, 1), "peak": BF16_MFMA_PEAK_TF,
                                    "unit": f"TFLOP/s ({what}; " +
                                            ("the centroid update inside the time)" if stage == "kmeans" else "residual packing inside the time)"),
                                    "frac": round(tf / BF16_MFMA_PEAK_TF, 4)}
    rec["input"] = "generated on the device chunk by chunk (synthetic.DeviceMixtureSource); no host copy of the embeddings exists"
    return index, rec


def sharded_index_build(torch, dist, rank, world, docs, kmeans_iters, dev):
    """BASELINE config 5's build half on the N ranks of this job: the passages sharded in contiguous ranges, one rank's
    range generated on its own GPU, sample / k-means / statistics / compress / IVF through
    distributed_index.index_device_sharded over the process group (RCCL: one all-gather of the [sums | counts] blocks
    per k-means iteration).  Seconds per stage = MAX over ranks; the centroids must agree on every rank."""
    from colbert_jl_amd import synthetic
    from colbert_jl_amd.distributed_index import index_device_sharded
    n_local = docs // world
    src = synthetic.DeviceMixtureSource(seed=610 + rank, n_docs=n_local, device=dev)
    index, rec = index_device_sharded(src, rank * n_local, n_local * world, nbits=2, kmeans_niters=kmeans_iters, seed=62)
    staged = dist.get_backend() != "nccl"
    cdev = torch.device("cpu") if staged else dev
    keys = ["sample_and_split_s", "kmeans_s", "codec_stats_s", "chunks_s", "build_ivf_s", "total_build_s"]
    t = torch.tensor([rec[kk] for kk in keys], dtype=torch.float64, device=cdev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    # identical centroids on every rank: min and max over the ranks of a checksum agree
    ck = index["centroids"].view(torch.int32).to(torch.int64).sum().reshape(1).to(cdev)
    lo, hi = ck.clone(), ck.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN); dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    out = {kk: round(float(v), 3) for kk, v in zip(keys, t.tolist())}
    out.update({"ranks": world, "passages_total": rec["passages_total"], "passages_per_rank": n_local, "K": rec["K"],
                "sample_points_total": rec["sample_points_total"], "kmeans_iters": rec["kmeans_iters"],
                "kmeans_s_per_iter": round(out["kmeans_s"] / max(rec["kmeans_iters"], 1), 4),
                "kmeans_exchange_bytes_per_rank_per_iter": rec["kmeans_exchange_bytes_per_rank_per_iter"],
                "centroids_identical_on_all_ranks": bool(lo.item() == hi.item()),
                "exchange": "staged through host memory (gloo rehearsal)" if staged else "RCCL all_gather_into_tensor",
                "note": "seconds per stage are the MAX over the ranks; every rank builds and keeps its own shard"})
    n_prod, what = nearest_products()
    tf = n_prod * 2.0 * 128 * rec["sample_points_total"] * rec["K"] * rec["kmeans_iters"] / max(out["kmeans_s"], 1e-9) / 1e12
    out["kmeans_roofline"] = {"bound": "mfma", "achieved": round(tf, 1), "peak": BF16_MFMA_PEAK_TF * world,
                              "unit": f"TFLOP/s over all ranks ({what}; exchange and update inside the time)",
                              "frac": round(tf / (BF16_MFMA_PEAK_TF * world), 4)}
    return out


def encoder_l2_operand_bytes(M, N, Kd, gemm="f16x3"):
    """Operand bytes the work-groups of one Linear layer pull through L2, mirroring linear_planes' tile choice
    (csrc/encoder.hip): every work-group streams the planes of its A rows and B rows of the K range it owns -- two
    16-bit planes per operand for f16x3 / bf16x3, three for bf16x6 (the fp32 MFMA mode reads fp32 operands)."""
    def wgs(bm, bn):
        return -(-N // bn) * -(-M // bm)
    bytes_per_el = {"f16x3": 4, "bf16x3": 4, "bf16x6": 6}.get(gemm, 4)
    if wgs(128, 128) >= 256:         # passage batches (a GELU epilogue keeps 128 x 128: not modelled here)
        wide = bytes_per_el == 4 and N % 4 == 0
        bm, bn, ks = (256, 256, 1) if wide and wgs(256, 256) >= 200 else (128, 256, 1) if wide and wgs(128, 256) >= 384 else (128, 128, 1)
    else:
        bm, bn, ks = 64, 64, 1
        min_slice = 192 if wgs(64, 64) < 64 else 384
        while ks < 8 and wgs(64, 64) * ks < 768 and Kd % (ks * 2 * 32) == 0 and Kd // (ks * 2) >= min_slice:
            ks *= 2
    return wgs(bm, bn) * (bm + bn) * Kd * bytes_per_el, (bm, bn, ks)


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
    ap.add_argument("--no-passage-encoder", action="store_true", help="skip the 64 x 300 passage batch of the encoder leg")
    ap.add_argument("--no-latency", action="store_true", help="skip the one-query-at-a-time latency loop (profiling runs)")
    ap.add_argument("--force-gather", action="store_true", help="exercise the all-gather + merge path even with one rank (testing)")
    ap.add_argument("--uniform-codes", action="store_true",
                    help="the HEADLINE corpus draws its centroid codes uniformly (worst case for the candidate count) instead of topically")
    ap.add_argument("--no-overlap", action="store_true", help="one compute stream: batches strictly one after the other")
    ap.add_argument("--in-flight", type=int, default=2, help="batches in flight (compute streams / workspace slots), 1..4")
    ap.add_argument("--no-encoder", action="store_true",
                    help="skip the second measurement with the query encoder (bert-base geometry) in front of the search")
    ap.add_argument("--min-seconds", type=float, default=0.5,
                    help="the timed region is repeated (whole multiples of --steps) until it lasts at least this long")
    ap.add_argument("--no-sub", action="store_true",
                    help="headline only: skip worst_case_uniform_codes, built_index and batch_sweep (profiling runs)")
    ap.add_argument("--built-index", action="store_true",
                    help="make the config-2 built index (100k passages through this repo's own index build) the HEADLINE workload")
    ap.add_argument("--built-docs", type=int, default=100_000, help="passages of the built_index workload (BASELINE config 2)")
    ap.add_argument("--built-kmeans-iters", type=int, default=20, help="k-means iterations of the built index (reference default)")
    ap.add_argument("--exchange", choices=["two-phase", "single"], default="two-phase",
                    help="N > 1: the exchange `value` is measured with -- two-phase (default: a small all-gather of every shard's k largest "
                         "approximate scores gives all shards the GLOBAL threshold, then one all-gather of the packed top-k) or single "
                         "(BASELINE north_star's wording: ONE all-gather of the per-shard top-k per batch); the other one is in the same line")
    ap.add_argument("--no-built-1m", action="store_true",
                    help="skip built_index_1M (the headline corpus size through this repo's own device-resident index build, ~40 s)")
    ap.add_argument("--built-1m-docs", type=int, default=0,
                    help="passages of the built_index_1M workload (0: 1 000 000 when --docs is at least that, else skipped; tests pass a small value)")
    ap.add_argument("--no-index-build", action="store_true", help="N > 1: skip the sharded index build over the process group")
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
    backend = None

    def first_contact(what, fn):
        """Run one step of the N-rank set-up; a failure names the rank and the step on stderr and ends THIS rank with code 3 at
        once (the launcher -- launch_ranks above, or torchrun -- then stops the others): a first run on new hardware that
        cannot form its communicator costs a minute and says why, instead of a timeout somewhere inside the timed region."""
        try:
            return fn()
        except BaseException as e_:     # noqa: BLE001 -- including RuntimeError from RCCL and KeyboardInterrupt from a watchdog
            print(f"[bench preflight] rank {rank}/{world} (device {local_rank}): {what} FAILED: {type(e_).__name__}: {e_}",
                  file=sys.stderr, flush=True)
            os._exit(3)
    if world > 1 or args.force_gather:
        import torch.distributed as dist
        # COLBERT_BENCH_BACKEND / COLBERT_BENCH_DEVICE: test hooks to run several ranks on ONE GPU over gloo
        # (RCCL refuses two ranks on the same device); the measured configuration is always nccl, one GPU per rank
        backend = os.environ.get("COLBERT_BENCH_BACKEND", "nccl")
        if "COLBERT_BENCH_DEVICE" in os.environ:
            local_rank = int(os.environ["COLBERT_BENCH_DEVICE"])
        kw = {}
        if "COLBERT_BENCH_INIT_FILE" in os.environ:     # started by launch_ranks: file rendezvous, no port to pick
            kw["init_method"] = "file://" + os.environ["COLBERT_BENCH_INIT_FILE"]
        else:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29511")
        import datetime
        kw["timeout"] = datetime.timedelta(seconds=int(os.environ.get("COLBERT_BENCH_COLLECTIVE_TIMEOUT_S", "300")))
        if backend == "nccl":
            first_contact("init_process_group(nccl = RCCL)", lambda: dist.init_process_group(
                backend="nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank), **kw))
        else:
            first_contact(f"init_process_group({backend})", lambda: dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw))
    if world != args.gpus:
        print(f"[bench] --gpus {args.gpus} but WORLD_SIZE={world}: refusing to report a number for a different job size",
              file=sys.stderr)
        sys.exit(2)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    ranks_seen = world
    if world > 1 or args.force_gather:
        # counted by the communicator that moves the data (RCCL when the backend is nccl): every rank adds one
        one_t = torch.ones(1, dtype=torch.int32, device=dev if backend == "nccl" else "cpu")
        first_contact("first all-reduce (rank count)", lambda: (dist.all_reduce(one_t, op=dist.ReduceOp.SUM), torch.cuda.synchronize()))
        ranks_seen = int(one_t.item())
        if ranks_seen != args.gpus and not args.force_gather:
            print(f"[bench] the communicator counts {ranks_seen} ranks, --gpus {args.gpus}", file=sys.stderr)
            sys.exit(2)

    # ---- this rank's passage shard (generated directly, identical to the same passages of the full index)
    T, B, k = 32, (args.batch if args.batch > 0 else min(32 * world, 256)), args.k
    n_blocks = 8
    assert n_blocks % world == 0, "shards are aligned to the 8 generation blocks: use 1, 2, 4 or 8 GPUs"
    per = n_blocks // world
    index_build = None
    t0 = time.time()
    if args.built_index:
        assert world == 1, "--built-index is a single-GPU workload (BASELINE config 2)"
        if args.built_docs > 200_000:      # the fp32 embeddings no longer fit a host buffer comfortably: device-resident build
            from colbert_jl_amd.indexer import index_to_host
            didx, index_build = build_device_index(torch, args.built_docs, args.built_kmeans_iters, dev)
            shard = index_to_host(didx)
            del didx
        else:
            shard, index_build = build_config2_index(clb, args.built_docs, args.built_kmeans_iters, local_rank)
        K = int(shard["centroids"].shape[1])
        n_docs_total = args.built_docs
    else:
        K = synthetic.num_partitions_for(args.docs, 80.0)
        shard = synthetic.make_index(seed=2024, n_docs=args.docs, K=K, n_blocks=n_blocks,
                                     blocks=range(rank * per, (rank + 1) * per), topical=not args.uniform_codes)
        n_docs_total = args.docs
    t_gen = time.time() - t0
    t0 = time.time()
    s = clb.Searcher(index=shard, device=local_rank, pid_offset=int(shard["pid_offset"]))
    if args.mode >= 0:
        s.set_mode(args.mode)
    if world > 1:
        from colbert_jl_amd.distributed import sync_bound_consts
        # one error bound on every shard (the threshold of the two-phase search is global)
        first_contact("all-reduce of the error-bound constants", lambda: sync_bound_consts(s))
    t_load = time.time() - t0
    # the timed region (warmup + steps batches) never issues a query twice; longer legs (sustained, sweeps) cycle the pool
    n_queries = max(B * 8, 768, -(-B * (args.steps + args.warmup + 1) // 256) * 256)      # 768 at the default 3 + 20 steps of 32: profiling runs (fewer steps) see the same queries
    if args.built_index:
        Q = synthetic.make_queries(shard, seed=77, n_queries=n_queries, T=T)
    else:
        Q = synthetic.make_topic_queries(shard["centroids"], seed=77, n_queries=n_queries, T=T)   # (dim, T, nq)
    Qdev = torch.from_numpy(np.ascontiguousarray(Q.transpose(2, 1, 0))).to(dev)               # (nq, T, dim)
    gather = world > 1 or args.force_gather
    # Two result buffers alternate so that the exchange of batch i (one RCCL all-gather of the packed per-shard
    # top-k + the merge kernel, on a side stream) overlaps the search of batch i+1 on the main stream.
    # Two batches in flight: batch i runs on compute stream i % NF with its own workspace slot and result buffers, so
    # the latency-bound kernels of one batch (selection, top-k: one work-group per query) overlap the scoring kernels
    # of the other.  --no-overlap puts every batch on one stream.
    NF = max(1, min(4, args.in_flight))
    compute = [torch.cuda.Stream(device=dev) for _ in range(NF)]
    comm = torch.cuda.Stream(device=dev) if gather else None
    # With several shards the search runs in two phases around a second, small all-gather (every shard's k largest
    # approximate scores): all shards then cut at the GLOBAL k-th score and re-score ~k/N passages each instead of
    # ~k.  COLBERT_BENCH_TWO_PHASE=0/1 overrides.
    two_phase_default = gather and s.mode == 1 and ((world >= 2 and args.exchange == "two-phase") if "COLBERT_BENCH_TWO_PHASE" not in os.environ
                                                    else os.environ["COLBERT_BENCH_TWO_PHASE"] == "1")
    import torch.distributed as _dist
    overlap = [not args.no_overlap]

    class Plan:
        """One batch size / exchange mode: its DeviceSearch objects (one per batch in flight), result buffers, and the
        step that runs batch i."""

        def __init__(self, Bp, two_phase):
            self.B, self.two_phase = Bp, two_phase
            self.runs = [DeviceSearch(s, T, Bp, k, args.nprobe, slot=i) for i in range(NF)]
            self.merged = [(torch.empty((Bp, k), dtype=torch.int64, device=dev), torch.empty((Bp, k), dtype=torch.float32, device=dev))
                           for _ in range(NF)]
            self.free_ev = [None] * NF          # buffer set i may be overwritten once its exchange has finished
            self.gath = [torch.empty((max(world, 1) * Bp, k), dtype=torch.float32, device=dev) for _ in range(NF)] if two_phase else None

        def search_shard(self, r, Qb, i):
            """This rank's part of one batch on the current stream (results in r.packed)."""
            if not self.two_phase:
                r(Qb)
                return
            main = torch.cuda.current_stream(dev)
            lt = r.phase1(Qb)
            e1 = torch.cuda.Event()
            e1.record(main)
            with torch.cuda.stream(comm):
                comm.wait_event(e1)
                _dist.all_gather_into_tensor(self.gath[i % NF], lt)
                e2 = torch.cuda.Event()
                e2.record(comm)
            main.wait_event(e2)
            r.phase2(Qb, self.gath[i % NF].view(world, self.B, k))

        def queries(self, i):
            off = (i * self.B) % (n_queries - self.B + 1)
            return Qdev[off:off + self.B]

        def step(self, i):
            with torch.cuda.stream(compute[i % NF] if overlap[0] else compute[0]):
                return self.step_on_current_stream(i, self.queries(i))

        def step_on_current_stream(self, i, Qb):
            r = self.runs[i % NF]
            if not gather:
                return r(Qb)
            main = torch.cuda.current_stream(dev)
            if self.free_ev[i % NF] is not None:
                main.wait_event(self.free_ev[i % NF])
            self.search_shard(r, Qb, i)
            done = torch.cuda.Event()
            done.record(main)
            with torch.cuda.stream(comm):
                comm.wait_event(done)
                g = all_gather_packed(r.packed)
                out = merge_packed(g, self.B, k, out_p=self.merged[i % NF][0], out_s=self.merged[i % NF][1])
                ev = torch.cuda.Event()
                ev.record(comm)
                self.free_ev[i % NF] = ev
            return out

    plan = Plan(B, two_phase_default)
    two_phase = two_phase_default
    run = plan.runs[0]

    def barrier():
        torch.cuda.synchronize()
        if gather:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(n_steps, first, pl=None):
        """n_steps steps between two barrier + synchronize pairs; MAX over ranks of the wall time."""
        pl = pl or plan
        barrier()
        t0 = time.perf_counter()
        for i in range(n_steps):
            pl.step(first + i)
        barrier()
        dt = time.perf_counter() - t0
        if world > 1:
            tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dt = float(tmax.item())
        return dt

    def sustained_of(pl, min_seconds):
        """The same loop repeated in whole multiples of --steps until the region lasts `min_seconds`; every rank uses the
        same count."""
        for i in range(args.warmup):
            pl.step(i)
        probe = timed(args.steps, args.warmup, pl)
        reps = max(1, int(np.ceil(min_seconds / max(probe, 1e-6))))
        if world > 1:
            r_t = torch.tensor([reps], dtype=torch.int64, device=dev)
            dist.all_reduce(r_t, op=dist.ReduceOp.MAX)
            reps = int(r_t.item())
        n = reps * args.steps
        return n, timed(n, args.warmup + args.steps, pl)

    # ---- N > 1, BEFORE any timing: one line per rank on stderr -- device, RCCL version, ranks seen, shard
    # sizes, centroid checksum -- then ONE batch through every shard's search, the exchange and the merge, checked on rank 0
    # against the CPU oracle's search of the UNSHARDED index.  Any failing collective, a rank whose replicated centroids differ,
    # shards that do not tile the corpus, or a merged result that is not the oracle's end the run with a non-zero code here.
    merged_check = None
    if gather:
        import zlib
        cen = np.ascontiguousarray(shard["centroids"])
        # COLBERT_BENCH_FAULT (tests/test_gpu_dist_search.py): "centroids:R" makes rank R report a different centroid checksum,
        # "collective:R" makes its first preflight collective raise -- the run must end at once with the rank's message
        fault = os.environ.get("COLBERT_BENCH_FAULT", "")
        if fault == f"centroids:{rank}":
            cen = cen + np.float32(1.0)
        me = {"rank": rank, "device": local_rank, "device_name": torch.cuda.get_device_name(local_rank), "backend": backend,
              "rccl": ".".join(str(v) for v in torch.cuda.nccl.version()) if backend == "nccl" else None,
              "ranks_seen": ranks_seen, "passages": int(shard["doclens"].size), "embeddings": int(shard["codes"].size),
              "pid_offset": int(shard["pid_offset"]), "centroids_crc32": f"{zlib.crc32(cen.tobytes()) & 0xffffffff:08x}",
              "exchange": "two-phase" if two_phase else "single", "batch": B}
        print("[bench preflight] " + json.dumps(me), file=sys.stderr, flush=True)
        everyone = [None] * world
        def gather_descriptions():
            if fault == f"collective:{rank}":
                raise RuntimeError("injected fault (COLBERT_BENCH_FAULT)")
            dist.all_gather_object(everyone, me)
        first_contact("all_gather_object of the rank descriptions", gather_descriptions)
        problems = []
        if len({e_["centroids_crc32"] for e_ in everyone}) != 1:
            problems.append("the replicated centroids differ between ranks: " + ", ".join(f"rank {e_['rank']}: {e_['centroids_crc32']}" for e_ in everyone))
        if world > 1:
            order = sorted(everyone, key=lambda e_: e_["pid_offset"])
            if order[0]["pid_offset"] != 0 or any(a["pid_offset"] + a["passages"] != b_["pid_offset"] for a, b_ in zip(order, order[1:])) \
                    or order[-1]["pid_offset"] + order[-1]["passages"] != n_docs_total:
                problems.append("the shards do not tile the corpus: " + ", ".join(f"rank {e_['rank']}: [{e_['pid_offset']}, +{e_['passages']})" for e_ in everyone))
            if len({e_["device"] for e_ in everyone}) != world and backend == "nccl":
                problems.append("two ranks share a device: " + ", ".join(f"rank {e_['rank']}: device {e_['device']}" for e_ in everyone))
        if problems:
            if rank == 0:
                for p_ in problems:
                    print("[bench preflight] FAILED: " + p_, file=sys.stderr, flush=True)
            os._exit(4)
        def one_batch():
            merged = plan.step(0)
            barrier()
            return merged
        mp_, ms_ = first_contact("one batch: search of every shard + exchange + merge", one_batch)
        verdict = torch.ones(1, dtype=torch.int32, device=dev if backend == "nccl" else "cpu")
        if rank == 0 and world > 1 and not args.no_cpu and not args.built_index:
            from oracle import oracle as orc
            orc.build()
            t1 = time.time()
            full = synthetic.make_index(seed=2024, n_docs=args.docs, K=K, n_blocks=n_blocks, blocks=range(n_blocks), topical=not args.uniform_codes)
            full["emb2pid"] = orc.build_emb2pid(full["doclens"])
            mp_h, ms_h = mp_.cpu().numpy(), ms_.cpu().numpy()
            n_chk = min(plan.B, 8 if args.docs <= 400_000 else 4)
            ok = True
            for b in range(n_chk):
                rp, rs, _ = orc.search(full, Q[:, :, b], args.nprobe, k)
                ok = ok and bool(np.array_equal(rp, mp_h[b])) and bool(np.array_equal(rs.view(np.uint32), ms_h[b].view(np.uint32)))
            merged_check = {"queries": n_chk, "merged_equals_oracle_on_the_unsharded_index": ok, "seconds": round(time.time() - t1, 1),
                            "when": "before any timing (preflight)",
                            "note": "pids identical and fp32 score bits identical, exchange = " + ("two-phase" if two_phase else "single")}
            del full
            print("[bench preflight] merged_vs_oracle " + json.dumps(merged_check), file=sys.stderr, flush=True)
            verdict[0] = 1 if ok else 0
        first_contact("broadcast of the preflight verdict", lambda: dist.broadcast(verdict, src=0))
        if int(verdict.item()) != 1:
            if rank == 0:
                print("[bench preflight] FAILED: the merged top-k of one batch is not the CPU oracle's on the unsharded index: no timing is reported",
                      file=sys.stderr, flush=True)
            os._exit(5)
        if rank == 0:
            print(f"[bench preflight] ok: {world} rank(s), {ranks_seen} seen by the communicator, centroids identical, shards tile {n_docs_total} passages"
                  + ("" if merged_check is None else f", merged result of one batch == oracle on {merged_check['queries']} queries"),
                  file=sys.stderr, flush=True)

    # (0) pre-conditioning + the sustained figure (a 20-step region is ~25 ms: too short to trust on its own, and a GPU
    #     that has just come out of index generation idles at a low clock)
    sustained_steps, sustained_s = sustained_of(plan, args.min_seconds)
    # (1) the contract's measurement on the warm device: --warmup untimed steps, then exactly --steps steps between
    #     barrier + synchronize pairs, per-kernel event timing OFF
    for i in range(args.warmup):
        plan.step(i)
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
        out = plan.step(i)
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
        plan.step(i)
    busy = result_of(probe_i)
    for i in range(probe_i + 1, probe_i + 1 + NF):
        plan.step(i)
    barrier()
    in_flight_ok = bool(torch.equal(alone[0], busy[0]) and torch.equal(alone[1], busy[1]))
    if world > 1:
        okt = torch.tensor([1 if in_flight_ok else 0], dtype=torch.int64, device=dev)
        dist.all_reduce(okt, op=dist.ReduceOp.MIN)
        in_flight_ok = bool(okt.item())

    # ---- N > 1: the same 32-query batch at every N (so that the N-GPU point and the 1-GPU point of a strong-scaling
    # curve run the same batch), and the single-exchange mode north_star describes (ONE all-gather of the per-shard top-k
    # per batch, every shard cutting at its own threshold) beside the two-phase default
    fixed32 = single_exchange = two_phase_leg = None
    if world > 1:
        if B != 32:
            p32 = Plan(32, two_phase_default)
            n32, s32 = sustained_of(p32, min(args.min_seconds, 0.3))
            fixed32 = {"batch": 32, "value": round(32 * n32 / s32, 2), "unit": "queries/s", "ms_per_step": round(s32 / n32 * 1e3, 4),
                       "steps": n32, "note": "the headline run of N = 1 uses this batch; value above uses 32 queries per GPU"}
        if two_phase_default:
            p1x = Plan(B, False)
            n1x, s1x = sustained_of(p1x, min(args.min_seconds, 0.3))
            single_exchange = {"batch": B, "value": round(B * n1x / s1x, 2), "unit": "queries/s",
                               "ms_per_step": round(s1x / n1x * 1e3, 4), "steps": n1x,
                               "note": "one all-gather (packed per-shard top-k) per batch; every shard re-scores against its own k-th approximate score"}
        elif s.mode == 1:
            p2x = Plan(B, True)
            n2x, s2x = sustained_of(p2x, min(args.min_seconds, 0.3))
            two_phase_leg = {"batch": B, "value": round(B * n2x / s2x, 2), "unit": "queries/s", "ms_per_step": round(s2x / n2x * 1e3, 4),
                             "steps": n2x, "note": "global-threshold exchange between the passes + one all-gather of the packed top-k"}

    # ---- the metric as the reference's search(::String) defines it (src/searching.jl:93-128): encode_queries first.
    # No checkpoint exists in the build image, so the encoder has bert-base-uncased GEOMETRY with random weights and
    # runs on synthetic token ids; random weights give meaningless embeddings that would change the candidate
    # statistics, so the search half of the step still consumes the synthetic queries of the headline line: the
    # step costs exactly "encode B queries + search B queries", back to back on one stream.  With N ranks every rank
    # encodes B / N of the batch's queries and one all-gather (B x 32 x 128 fp32) gives every shard all of them.
    e2e = None
    if not args.no_encoder:
        from colbert_jl_amd.encoder import BERT_BASE, random_weights
        enc = clb.BertEncoder(random_weights(BERT_BASE, 128, seed=5), dict(BERT_BASE), dim=128, device=local_rank)
        rng = np.random.default_rng(6)
        d_ids = torch.from_numpy(rng.integers(1, BERT_BASE["vocab_size"] + 1, size=(n_queries, T)).astype(np.int32)).to(dev)
        d_mask = torch.ones((n_queries, T), dtype=torch.uint8, device=dev)
        d_skip = torch.tensor([1], dtype=torch.int64, device=dev)
        split = world > 1 and B % world == 0 and backend == "nccl"
        Bl = B // world if split else B                      # queries this rank encodes
        q_encs = [torch.empty((B, T, 128), dtype=torch.float32, device=dev) for _ in range(NF)]
        q_part = [torch.empty((Bl, T, 128), dtype=torch.float32, device=dev) for _ in range(NF)] if split else q_encs
        enc_done = [None]          # the encoder has ONE activation workspace: its calls are chained by an event
        # COLBERT_BENCH_ENCODER_GRAPH=1: the encoder's ~90 launches per batch as ONE captured HIP graph per result buffer
        # (static id buffers; the step copies its batch's ids in first)
        enc_graphs = None
        if os.environ.get("COLBERT_BENCH_ENCODER_GRAPH", "0") != "0":       # measured: 1.39 against 1.38 ms per batch -- no gain, off
            ids_static = [torch.empty((Bl, T), dtype=torch.int32, device=dev) for _ in range(NF)]
            for j in range(NF):
                ids_static[j].copy_(d_ids[0:Bl])
            mask_static = torch.ones((Bl, T), dtype=torch.uint8, device=dev)
            torch.cuda.synchronize()
            enc_graphs = [enc.capture_query_graph(ids_static[j], mask_static, d_skip, q_part[j]) for j in range(NF)]

        def encode_into(i, st):
            off = (i * B) % (n_queries - B + 1) + (rank * Bl if split else 0)
            if enc_done[0] is not None:
                st.wait_event(enc_done[0])
            if enc_graphs is not None:
                ids_static[i % NF].copy_(d_ids[off:off + Bl])
                enc_graphs[i % NF].replay()
            else:
                enc.query_embeddings_device(d_ids[off:off + Bl], d_mask[off:off + Bl], d_skip, q_part[i % NF])
            ev = torch.cuda.Event()
            ev.record(st)
            enc_done[0] = ev
            if split:
                _dist.all_gather_into_tensor(q_encs[i % NF], q_part[i % NF])

        def step_e2e(i):
            st = compute[i % NF] if overlap[0] else compute[0]
            with torch.cuda.stream(st):
                encode_into(i, st)
                return plan.step_on_current_stream(i, plan.queries(i))

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
            with torch.cuda.stream(compute[0]):
                encode_into(i, compute[0])
        barrier()
        dt_enc = time.perf_counter() - t0
        if world > 1:
            tm = torch.tensor([dt, dt_enc], dtype=torch.float64, device=dev)
            dist.all_reduce(tm, op=dist.ReduceOp.MAX)
            dt, dt_enc = float(tm[0].item()), float(tm[1].item())
        # per-stage HIP-event times of the encoder (on the launching stream) -> roofline of its dominant Linear layer
        enc.profile_enable(True)
        n_prof = 5
        for i in range(n_prof):
            with torch.cuda.stream(compute[0]):
                enc.query_embeddings_device(d_ids[0:Bl], d_mask[0:Bl], d_skip, q_part[0])
        torch.cuda.synchronize()
        eprof = enc.profile_read()
        enc.profile_enable(False)
        enc_roof = None
        if eprof:
            H, I, Lyr = BERT_BASE["hidden_size"], BERT_BASE["intermediate_size"], BERT_BASE["num_hidden_layers"]
            M = Bl * T
            shapes = {"linear_qkv": (3 * H, H), "linear_attn_out_ln": (H, H), "linear_ffn_in_gelu": (I, H),
                      "linear_ffn_out_ln": (H, I), "linear_projection": (128, H)}
            lin = {kname: v for kname, v in eprof.items() if kname in shapes}
            domk = max(lin.items(), key=lambda kv: kv[1]["ms"])[0]
            nprod = {"bf16x6": 6, "bf16x3": 3, "f16x3": 3}.get(enc.gemm, 1)
            Nn, Kd = shapes[domk]
            ms1 = eprof[domk]["ms"] / max(eprof[domk]["launches"], 1)
            peak = BF16_MFMA_PEAK_TF if nprod > 1 else F32_MFMA_PEAK_TF
            ach = nprod * 2.0 * M * Nn * Kd / (ms1 * 1e-3) / 1e12
            l2b, tile = encoder_l2_operand_bytes(M, Nn, Kd, enc.gemm)
            op16 = "fp16" if enc.gemm == "f16x3" else "bf16"
            enc_roof = {"kernel": f"{domk} ({M} x {Nn} x {Kd}, gemm_planes2_kernel {tile[0]}x{tile[1]} tiles, split-K {tile[2]})",
                        "bound": "mfma", "achieved": round(ach, 1), "peak": peak,
                        "unit": f"TFLOP/s ({op16 + ', ' + str(nprod) + ' products per fp32 product' if nprod > 1 else 'fp32 MFMA'})",
                        "frac": round(ach / peak, 4), "ms_per_launch": round(ms1, 4),
                        "fp32_equivalent_TFLOPs": round(2.0 * M * Nn * Kd / (ms1 * 1e-3) / 1e12, 1),
                        "l2_operand_bytes": int(l2b), "l2_operand_GBps": round(l2b / (ms1 * 1e-3) / 1e9, 1),
                        "note": "l2_operand_bytes = bytes of pre-split 16-bit operand planes the work-groups stream through L2 (tile rule of "
                                "csrc/encoder.hip); an XCD's L2 delivers ~65 GB/s to one CU, which together with MFMA issue bounds these "
                                "small-M GEMMs (tools/microbench/gemm_planes_bench.hip); kernel statistics: profiles/r04_encoder_kernel_stats.csv",
                        "all_stages_ms_per_encode": {kn: round(v["ms"] / n_prof, 4) for kn, v in eprof.items()}}
        e2e = {"value": round(B * args.steps / dt, 2), "unit": "queries/s", "ms_per_step": round(dt / args.steps * 1e3, 4),
               "encoder_ms_per_step": round(dt_enc / args.steps * 1e3, 4),
               "dtype": "f32" if enc.gemm == "f32" else (f"f32 ({enc.gemm}: every fp32 product of the Linear layers as exact "
                        f"{'fp16' if enc.gemm == 'f16x3' else 'bf16'} MFMA products of operands pre-split into 16-bit planes, fp32 accumulation)"),
               "encoder": "bert-base-uncased geometry (12 x 768, 12 heads, FFN 3072) + Dense 768->128, random weights, "
                          "synthetic token ids; " + (f"every rank encodes {Bl} of the batch's {B} queries, one all-gather of Q"
                                                     if split else "this rank encodes the whole batch"),
               "roofline": enc_roof,
               "encoder_launches": "one captured HIP graph per batch" if enc_graphs is not None else "stream launches",
               "note": "encode_queries + search per step; the search consumes the synthetic queries of the headline line"}
        # the same leg at larger batches (N = 1): the query encoder's Linear layers at M = 32 B rows are far from filling the chip at
        # B = 32, so a throughput-oriented deployment would batch more queries per encode -- reported beside the
        # contract's B = 32 figure, never instead of it
        if world == 1 and not args.no_sub:
            e2e["larger_batches"] = {}
            for Bs in (64, 128):
                if Bs <= B or Bs > n_queries:
                    continue                                             # (this leg cycles the query pool, like the sweeps)
                pl = Plan(Bs, False)
                qb = [torch.empty((Bs, T, 128), dtype=torch.float32, device=dev) for _ in range(NF)]
                prev = [None]

                def step_big(i, pl=pl, qb=qb, prev=prev, Bs=Bs):
                    st = compute[i % NF] if overlap[0] else compute[0]
                    off = (i * Bs) % (n_queries - Bs + 1)
                    with torch.cuda.stream(st):
                        if prev[0] is not None:
                            st.wait_event(prev[0])                       # ONE activation workspace: encodes are chained
                        enc.query_embeddings_device(d_ids[off:off + Bs], d_mask[off:off + Bs], d_skip, qb[i % NF])
                        ev = torch.cuda.Event(); ev.record(st); prev[0] = ev
                        return pl.step_on_current_stream(i, pl.queries(i))
                for i in range(3):
                    step_big(i)
                barrier()
                t0 = time.perf_counter()
                for i in range(args.steps):
                    step_big(3 + i)
                barrier()
                dtb = time.perf_counter() - t0
                t0 = time.perf_counter()
                for i in range(args.steps):
                    with torch.cuda.stream(compute[0]):
                        enc.query_embeddings_device(d_ids[0:Bs], d_mask[0:Bs], d_skip, qb[0])
                barrier()
                dte = time.perf_counter() - t0
                e2e["larger_batches"][str(Bs)] = {"value": round(Bs * args.steps / dtb, 2), "ms_per_step": round(dtb / args.steps * 1e3, 4),
                                                  "encoder_ms_per_step": round(dte / args.steps * 1e3, 4)}
                del pl, qb
        # The serving shape (searcher.TextSearch.search_many does the same for texts): ONE encode of 128 queries
        # on the encoder's own stream, its four 32-query search batches on the compute streams behind it, and the NEXT encode
        # enqueued behind the first of those batches -- the encoder's Linear layers fill the chip at 4 096 rows (0.34 of the
        # matrix peak against 0.18 at 1 024), the search keeps its 32-query batches, and an encode runs beside the three
        # remaining search batches of the group before it.  Latency of a query = from its group's encode entering the stream to
        # the end of its own search batch (HIP events), under this sustained load.
        if world == 1 and not args.no_sub and n_queries >= 256 and B == 32:
            Be = 128
            enc_stream = torch.cuda.Stream(device=dev)
            qenc = [torch.empty((Be, T, 128), dtype=torch.float32, device=dev) for _ in range(2)]

            def serve(Bs, gated, n_groups):
                """n_groups groups of Be queries: one encode each, Be / Bs search batches; -> (seconds, per-batch latencies in ms)"""
                pl = plan if Bs == B else Plan(Bs, False)
                per = Be // Bs
                gate = [None]             # the first search batch of the previous group has finished: the next encode may start
                lat = []

                def group(g, record):
                    off = (g * Be) % (n_queries - Be + 1)
                    with torch.cuda.stream(enc_stream):
                        if gated and gate[0] is not None:
                            enc_stream.wait_event(gate[0])
                        t_in = torch.cuda.Event(enable_timing=True)
                        t_in.record(enc_stream)
                        enc.query_embeddings_device(d_ids[off:off + Be], d_mask[off:off + Be], d_skip, qenc[g % 2])
                        encoded = torch.cuda.Event()
                        encoded.record(enc_stream)
                    for j in range(per):
                        i = per * g + j
                        st = compute[i % NF] if overlap[0] else compute[0]
                        with torch.cuda.stream(st):
                            st.wait_event(encoded)
                            pl.step_on_current_stream(i, pl.queries(i))
                            t_out = torch.cuda.Event(enable_timing=True)
                            t_out.record(st)
                            if j == 0:
                                gate[0] = t_out
                            if record:
                                lat.append((t_in, t_out))
                for g in range(2):
                    group(g, False)
                barrier()
                t0 = time.perf_counter()
                for g in range(n_groups):
                    group(2 + g, True)
                barrier()
                dts = time.perf_counter() - t0
                return dts, np.array([a.elapsed_time(b_) for a, b_ in lat])

            n_groups = max(args.steps // 2, 8)
            shapes = {}
            for Bs, gated in ((32, True), (32, False), (64, True), (64, False)):
                if Bs > n_queries // 4:
                    continue
                dts, lat_ms = serve(Bs, gated, n_groups)
                shapes[f"search_batch_{Bs}_{'gated' if gated else 'free'}"] = {
                    "value": round(Be * n_groups / dts, 2), "ms_per_group": round(dts / n_groups * 1e3, 4),
                    "latency_ms_per_query": {"p50": round(float(np.quantile(lat_ms, 0.5)), 4), "p99": round(float(np.quantile(lat_ms, 0.99)), 4)}}
            main_shape = shapes["search_batch_32_gated"]
            e2e["serving_shape"] = {
                "value": main_shape["value"], "unit": "queries/s", "queries_per_encode": Be, "queries_per_search_batch": B,
                "groups": n_groups, "ms_per_group": main_shape["ms_per_group"], "latency_ms_per_query": main_shape["latency_ms_per_query"],
                "variants": shapes,
                "note": "value: one encode of 128 queries, four 32-query search batches behind it, the next encode gated on the first of them; "
                        "latency = the group's encode entering its stream -> end of the query's own search batch (HIP events) under sustained "
                        "load, groups enqueued back to back; variants: 64-query search batches / the next encode not gated"}
            del qenc
        # the passage side of the same encoder (what index() spends its time in: 1 M passages = 15 600 such batches): one batch
        # of index_bsize = 64 passages x doc_maxlen = 300 tokens, output left on the device (clb_encode_docs_device)
        if rank == 0 and not args.no_passage_encoder:
            Np, Lp = 64, 300
            gen = torch.Generator(device=dev)
            gen.manual_seed(5)
            p_ids = torch.randint(1, BERT_BASE["vocab_size"] + 1, (Np, Lp), generator=gen, device=dev, dtype=torch.int32)
            p_mask = torch.ones((Np, Lp), dtype=torch.uint8, device=dev)
            p_skip = torch.tensor([1, 1013, 1014], dtype=torch.int64, device=dev)
            for _ in range(2):
                enc.doc_embeddings_device(p_ids, p_mask, p_skip)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(5):
                enc.doc_embeddings_device(p_ids, p_mask, p_skip)
            torch.cuda.synchronize()
            dtp = (time.perf_counter() - t0) / 5
            enc.profile_enable(True)
            for _ in range(3):
                enc.doc_embeddings_device(p_ids, p_mask, p_skip)
            torch.cuda.synchronize()
            pprof = enc.profile_read()
            enc.profile_enable(False)
            # what index() really feeds the encoder: passages of ~86 tokens back to back WITHOUT padding rows
            # (clb_encode_docs_packed_device; tensorize_docs would pad every batch of 64 to its longest passage, ~160 tokens), as
            # many as fit the row budget of indexer.EncoderSource (170 x 256 rows: whole rounds of tiles for every Linear)
            prng = np.random.default_rng(6)
            plens = np.clip(np.rint(86 + 30 * prng.standard_normal(700)), 8, 299).astype(np.int32)
            plens = plens[:int(np.searchsorted(np.cumsum(plens), 170 * 256, side="right"))]
            n_pk = int(plens.size)
            prow = int(plens.sum())
            pbuf = np.concatenate([prng.integers(1000, BERT_BASE["vocab_size"], size=prow).astype(np.int32),
                                   np.concatenate([np.arange(n, dtype=np.int32) for n in plens]),
                                   np.repeat(np.arange(plens.size, dtype=np.int32), plens),
                                   np.concatenate([[0], np.cumsum(plens)]).astype(np.int32)])
            dpk = torch.from_numpy(pbuf).to(dev)
            pk_args = (dpk[:prow], dpk[prow:2 * prow], dpk[2 * prow:3 * prow], dpk[3 * prow:], int(plens.max()), p_skip)
            for _ in range(2):
                enc.doc_embeddings_packed_device(*pk_args)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(5):
                enc.doc_embeddings_packed_device(*pk_args, n_out=prow)
            torch.cuda.synchronize()
            dpk_t = (time.perf_counter() - t0) / 5
            packed_leg = {"batch": f"{n_pk} passages, {prow} tokens (mean {prow / n_pk:.1f}), no padding rows", "ms_per_batch": round(dpk_t * 1e3, 3),
                          "passages_per_s": round(n_pk / dpk_t, 1), "tokens_per_s": round(prow / dpk_t, 1),
                          "encode_1M_passages_s": round(1e6 / (n_pk / dpk_t), 1)}
            Mp = Np * Lp
            H, I, Lyr = BERT_BASE["hidden_size"], BERT_BASE["intermediate_size"], BERT_BASE["num_hidden_layers"]
            shapes = {"linear_qkv": (3 * H, H), "linear_attn_out_ln": (H, H), "linear_ffn_in_gelu": (I, H), "linear_ffn_out_ln": (H, I)}
            nprod = {"bf16x6": 6, "bf16x3": 3, "f16x3": 3}.get(enc.gemm, 1)
            domk = "linear_ffn_in_gelu"               # the widest product; its stage holds nothing but the GEMM
            msl = pprof[domk]["ms"] / max(pprof[domk]["launches"], 1)
            ach = nprod * 2.0 * Mp * shapes[domk][0] * shapes[domk][1] / (msl * 1e-3) / 1e12
            lin_flop = 2.0 * Mp * Lyr * sum(a * b for a, b in shapes.values())
            e2e["passage_encoder"] = {
                "batch": f"{Np} passages x {Lp} tokens", "ms_per_batch": round(dtp * 1e3, 3), "passages_per_s": round(Np / dtp, 1),
                "fp32_equivalent_TFLOPs_linear_layers": round(lin_flop / dtp / 1e12, 1),
                "dominant_linear": {"kernel": f"{domk} ({Mp} x {shapes[domk][0]} x {shapes[domk][1]})", "bound": "mfma", "achieved": round(ach, 1),
                                    "peak": BF16_MFMA_PEAK_TF if nprod > 1 else F32_MFMA_PEAK_TF, "unit": "TFLOP/s",
                                    "frac": round(ach / (BF16_MFMA_PEAK_TF if nprod > 1 else F32_MFMA_PEAK_TF), 4), "ms_per_launch": round(msl, 4)},
                "stages_ms_per_batch": {kn: round(v["ms"] / 3, 4) for kn, v in pprof.items()},
                "encode_1M_passages_s": round(1e6 / (Np / dtp), 1),
                "packed_real_lengths": packed_leg,
                "note": "random token ids, every position attended (the worst case: real passages average ~80 of 300 positions)"}
        enc.close()

    # ---- work counters of one batch (for the roofline) and p50 latency, outside the timed region
    s.profile_enable(True, counters=True)
    plan.step(args.warmup)
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
    # the same single query as a captured HIP graph over a static query buffer (what a serving loop whose encoder
    # writes into that buffer would replay); one GPU only
    p50_graph_ms = None
    if lat and not gather and not two_phase:
        q_static = Qdev[0:1].clone()
        graph = one.capture(q_static)
        glat = []
        for i in range(40):
            q_static.copy_(Qdev[i:i + 1])
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            graph.replay()
            torch.cuda.synchronize()
            glat.append(time.perf_counter() - t1)
        p50_graph_ms = float(np.median(glat[5:]) * 1e3)
        del graph

    # ---- the reference API's own operating point: search(searcher, query::String, k) (src/searching.jl:93-128), ONE text
    # query per call -- tokenize (host, WordPiece over a synthetic bert-base-sized vocabulary) -> ids up -> encode 1 x 32
    # (bert-base geometry, random weights) -> search -> k (pid, score) pairs down; as stream launches and as ONE captured
    # HIP graph of encoder + search (Searcher.text_search).  N = 1 only.
    text_lat = None
    if lat and not gather and not args.no_encoder:
        import tempfile
        from colbert_jl_amd import tokenization
        from colbert_jl_amd.encoder import BERT_BASE, random_weights
        words = [f"w{i}" for i in range(BERT_BASE["vocab_size"] - 1000)]
        vocab = (["[PAD]"] + [f"[unused{i}]" for i in range(99)] + ["[UNK]", "[CLS]", "[SEP]", "[MASK]"] +
                 [f"[unused{i}]" for i in range(99, 99 + 896 - 0)])[:1000] + words
        with tempfile.NamedTemporaryFile("w", suffix=".txt", delete=False) as vf:
            vf.write("\n".join(vocab) + "\n")
        tokz = tokenization.WordPieceTokenizer(vf.name)
        os.unlink(vf.name)
        enc1 = clb.BertEncoder(random_weights(BERT_BASE, 128, seed=5), dict(BERT_BASE), dim=128, device=local_rank, tokenizer=tokz,
                               config=clb.ColBERTConfig(query_maxlen=T, dim=128))
        s.encoder = enc1
        rngq = np.random.default_rng(9)
        texts = [" ".join(words[j] for j in rngq.integers(0, len(words), size=int(rngq.integers(4, 12)))) for _ in range(48)]
        text_lat = {}
        results = {}
        for label, use_graph in (("stream_launches", False), ("hip_graph", True)):
            try:
                ts = s.text_search(k, args.nprobe, graph=use_graph)
                tl, tok_s = [], []
                for q in texts:
                    t1 = time.perf_counter()
                    out_q = ts(q)
                    tl.append(time.perf_counter() - t1)
                results[label] = out_q
                for q in texts[:16]:
                    t1 = time.perf_counter()
                    tokenization.tensorize_queries("[Q]", True, tokz, [q], T)
                    tok_s.append(time.perf_counter() - t1)
                text_lat[label] = {"p50_ms": round(float(np.median(tl[8:]) * 1e3), 4), "p90_ms": round(float(np.quantile(tl[8:], 0.9) * 1e3), 4),
                                   "tokenize_ms": round(float(np.median(tok_s) * 1e3), 4)}
                ts.close()
            except clb.ColBERTError as e_:      # e.g. fewer than k candidates for a random-weight query on a tiny corpus
                text_lat[label] = {"error": str(e_)[:200]}
        if len(results) == 2:
            text_lat["graph_matches_stream"] = bool(np.array_equal(results["stream_launches"][0], results["hip_graph"][0]) and
                                                    np.array_equal(results["stream_launches"][1], results["hip_graph"][1]))
        text_lat["note"] = ("search(searcher, query::String, k) per call: tokenizer + H2D of 32 ids + encoder (1 x 32 tokens, bert-base "
                            "geometry, random weights: the embeddings are meaningless, the work is not) + search + D2H of the top-k")
        s.encoder = None
        enc1.close()

    # ---- roofline of the dominant kernel (per launch = one batch on this rank's shard)
    dom = max(prof.items(), key=lambda kv: kv[1]["ms"])[0] if prof else None
    roof = None
    if dom:
        copy_gbs = measured_copy_rate(clb, local_rank)
        roof = roofline_of(dom, prof, stats, s.mode, T, K, B, copy_gbs=copy_gbs)
        # HBM bytes per launch: PMC counters cannot be read from inside this process; they come from the committed
        # rocprofv3 --pmc passes of this same command (tools/pmc_summary.py), which record the hash of the kernel
        # sources they were measured on.  A summary taken on different sources is stale: traffic stays null.
        roof["traffic"] = None
        pmc_file = os.path.join(ROOT, "profiles", "pmc_summary.json")
        # ... and on THIS workload (tools/gpu_profile.sh profiles the default one): bytes per launch do not transfer
        default_workload = (args.docs == 1_000_000 and not args.uniform_codes and not args.built_index and B == 32
                            and k == 1000 and args.nprobe == 2 and args.mode < 0)
        if world == 1 and not default_workload:
            roof["traffic_source"] = "null: profiles/pmc_summary.json was measured on the default workload, not this one"
        elif world == 1:
            traffic_from_pmc(pmc_file, dom, stats, roof)
        roof["all_kernels_ms_per_step"] = {kname: round(v["ms"] / max(prof_steps, 1), 4) for kname, v in prof.items()}
        roof["other_kernels"] = [roofline_of(kn, prof, stats, s.mode, T, K, B, copy_gbs=copy_gbs) for kn in ("score_approx", "score_exact", "centroid_scores")
                                 if kn in prof and kn != dom and prof[kn]["launches"]]
        # MFMA utilisation of the kernels (north_star: "MFMA utilisation on the scorer against the chip's peak"): cycles the matrix
        # pipes were busy over the cycles of the launch, from the committed PMC summary of this command (same hash gate as `traffic`)
        if world == 1 and default_workload and os.path.exists(pmc_file):
            from tools.pmc_summary import csrc_hash
            pmc_all = json.load(open(pmc_file))
            if pmc_all.get("csrc_sha256") == csrc_hash():
                for rec in [roof] + roof["other_kernels"]:
                    pk = pmc_all.get("kernels", {}).get(rec["kernel"], {})
                    if "mfma_pipe_busy_frac" in pk:
                        rec["mfma_pipe_busy_frac"] = pk["mfma_pipe_busy_frac"]
                        rec["effective_clock_GHz"] = pk.get("effective_clock_GHz")

    # ---- CPU baseline: the oracle (a port of the reference algorithm) on the host cores, rank 0, N = 1
    cpu = None
    if world == 1 and not args.no_cpu:
        from oracle import oracle as orc
        orc.build()
        emb2pid = orc.build_emb2pid(shard["doclens"])
        idx = dict(shard, emb2pid=emb2pid)
        nq_cpu, t_cpu, ok = 0, 0.0, True
        plan.search_shard(run, Qdev[0:B], 0); p, sc = run.out_p, run.out_s; torch.cuda.synchronize()
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

    # ---- N = 1 sub-records: the batch sizes the N-GPU runs use, the worst-case code distribution, and BASELINE config 2
    batch_sweep = worst = built = built_1m = None
    device_bytes = s.device_bytes
    if world > 1 and not args.no_index_build and not args.built_index:
        try:
            index_build = sharded_index_build(torch, dist, rank, world, args.docs, args.built_kmeans_iters, dev)
        except Exception as e:      # the search figures of the line stand on their own
            index_build = {"error": f"{type(e).__name__}: {e}"}
    if world == 1 and not args.no_sub and not args.force_gather:
        batch_sweep = {}
        for Bs in (64, 128, 256):
            if Bs == B or Bs > n_queries:
                continue
            pl = Plan(Bs, False)
            n_s, s_s = sustained_of(pl, 0.25)
            batch_sweep[str(Bs)] = {"value": round(Bs * n_s / s_s, 2), "ms_per_step": round(s_s / n_s * 1e3, 4), "steps": n_s}
            del pl
        s.close()
        s = None
        del shard
        if not args.uniform_codes and not args.built_index:
            t0 = time.time()
            u_idx = synthetic.make_index(seed=2024, n_docs=args.docs, K=K, n_blocks=n_blocks, blocks=range(n_blocks), topical=False)
            tg = time.time() - t0
            su = clb.Searcher(index=u_idx, device=local_rank, pid_offset=0)
            if args.mode >= 0:
                su.set_mode(args.mode)
            worst = measure_sub(torch, clb, su, u_idx, Q, 32, k, args.nprobe, args.steps, 0.4,
                                0 if args.no_cpu else 4, dev, in_flight=NF, T=T, pmc_key="uniform_codes" if args.docs == 1_000_000 else None)
            worst["workload"] = (f"synthetic {args.docs} passages (dim 128, nbits 2, doclen~80, K={K}, UNIFORM centroid codes: no "
                                 f"id-adjacent codes, the largest candidate sets), top-{k}, nprobe {args.nprobe}, batch 32")
            worst["setup_seconds"] = {"generate": round(tg, 1)}
            su.close()
            del u_idx
        if not args.built_index:
            bidx, brec = build_config2_index(clb, args.built_docs, args.built_kmeans_iters, local_rank)
            t0 = time.time()
            sb = clb.Searcher(index=bidx, device=local_rank, pid_offset=0)
            brec["upload_and_searcher_s"] = round(time.time() - t0, 2)
            if args.mode >= 0:
                sb.set_mode(args.mode)
            Qb = synthetic.make_queries(bidx, seed=78, n_queries=256, T=T)
            built = measure_sub(torch, clb, sb, bidx, Qb, 32, k, args.nprobe, args.steps, 0.4,
                                0 if args.no_cpu else 8, dev, in_flight=NF, T=T,
                                pmc_key="built_index" if (args.built_docs, args.built_kmeans_iters) == (100_000, 20) else None)
            built["workload"] = (f"BASELINE config 2: synthetic {args.built_docs} passages of 4096-component mixture embeddings "
                                 f"(dim 128, doclen~80) indexed by this repo's own build (k-means K={brec['K']}, nbits 2), "
                                 f"top-{k}, nprobe {args.nprobe}, batch 32; queries = noisy decompressed passage tokens")
            built["index_build"] = brec
            sb.close()
            del bidx
        n_built_1m = args.built_1m_docs or (1_000_000 if args.docs >= 1_000_000 else 0)
        if not args.built_index and not args.no_built_1m and n_built_1m > 0:
            # the headline corpus SIZE through this repo's own build: K = 131 072 k-means on the reference-sized sample,
            # 80 M embeddings compressed, IVF over 80 M codes -- all in HBM -- then searched like the other workloads
            from colbert_jl_amd.indexer import index_to_host
            didx, drec = build_device_index(torch, n_built_1m, args.built_kmeans_iters, dev)
            t0 = time.time()
            sd = clb.Searcher(index=didx)
            torch.cuda.synchronize()
            drec["searcher_from_device_arrays_s"] = round(time.time() - t0, 2)
            hidx = index_to_host(didx)
            del didx
            if args.mode >= 0:
                sd.set_mode(args.mode)
            Qd = synthetic.make_queries(hidx, seed=79, n_queries=256, T=T)
            built_1m = measure_sub(torch, clb, sd, hidx, Qd, 32, k, args.nprobe, args.steps, 0.4,
                                   0 if args.no_cpu else 8, dev, in_flight=NF, T=T,
                                   pmc_key="built_index_1M" if (n_built_1m, args.built_kmeans_iters) == (1_000_000, 20) else None)
            built_1m["workload"] = (f"synthetic {n_built_1m} passages of 4096-component mixture embeddings (dim 128, doclen~80) indexed by "
                                    f"this repo's own device-resident build (k-means K={drec['K']} on {drec['sample_points']} sample points, "
                                    f"nbits 2), top-{k}, nprobe {args.nprobe}, batch 32; queries = noisy decompressed passage tokens")
            built_1m["pass1_gather"] = {"form": ["vgpr", "lds-dma"][sd.pass1_gather[0]], "code_adjacency": round(sd.pass1_gather[1], 4)}
            built_1m["index_build"] = drec
            sd.close()
            del hidx

    if rank == 0:
        out = {"metric": "queries/sec, top-1000 on 1M-passage corpus", "value": round(qps, 2), "unit": "queries/s",
               "n_gpus": world, "ranks_seen": ranks_seen, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "strong",
               "vs_baseline": None, "dtype": "f32", "data": "synthetic",
               "config": {"workload": (f"BASELINE config 2: {n_docs_total} passages of mixture embeddings indexed by this repo's own build "
                                       f"(K={K}), " if args.built_index else
                                       f"synthetic {args.docs} passages (dim 128, nbits 2, doclen~80, K={K}"
                                       f"{', uniform codes' if args.uniform_codes else ''}), ") +
                                      f"top-{k}, nprobe {args.nprobe}, query_maxlen {T}, batch {B} queries/step, "
                                      f"passages sharded over {world} GPU(s)",
                          "search_mode": ("two-pass (fp16 MFMA prefilter + exact fp32 re-score)" if (prof and "score_approx" in prof) else "exact fp32 single pass")
                                         + (", global threshold exchange between the passes" if two_phase else ""),
                          # scalars a reader of the first 2 KB needs (the full records follow further down the line)
                          "batch": B,
                          "exchange": (None if not gather else "two-phase" if two_phase else "single"),
                          "p50_latency_ms": None if p50_ms is None else round(p50_ms, 4),
                          "p50_text_to_topk_ms": (text_lat or {}).get("stream_launches", {}).get("p50_ms"),
                          "end_to_end_with_query_encoder_qps": (e2e or {}).get("value"),
                          "end_to_end_serving_shape_qps": ((e2e or {}).get("serving_shape") or {}).get("value"),
                          "built_index_1M_qps": (built_1m or {}).get("value"),
                          "uniform_codes_qps": (worst or {}).get("value"),
                          "fixed_batch_32_qps": (fixed32 or {}).get("value"),
                          "single_exchange_qps": (single_exchange or {}).get("value")},
               "batch_policy": (f"fixed --batch {B}" if args.batch > 0 else
                                f"{B} queries per step = 32 per GPU (at most 256); the corpus is fixed (strong scaling of the passages), "
                                "fixed_batch_32 / batch_sweep give the same batch at every N"),
               "query_pool": {"distinct_queries": int(n_queries), "issued_in_timed_region": int(B * args.steps),
                              "note": "warmup + timed steps never issue a query twice; the sustained leg and the sweeps cycle the pool"},
               "batches_in_flight": NF if overlap[0] else 1, "in_flight_matches_serial": in_flight_ok,
               "one_batch_at_a_time": {"value": round(B * args.steps / serial_s, 2), "ms_per_step": round(serial_s / args.steps * 1e3, 4)},
               "sustained": {"steps": sustained_steps, "seconds": round(sustained_s, 4),
                             "value": round(B * sustained_steps / sustained_s, 2),
                             "note": "the same loop repeated until the timed region lasts --min-seconds"},
               "end_to_end_with_query_encoder": e2e,
               "p50_latency_ms": None if p50_ms is None else round(p50_ms, 4),
               "p50_latency_graph_replay_ms": None if p50_graph_ms is None else round(p50_graph_ms, 4),
               "p50_text_to_topk_ms": (text_lat or {}).get("stream_launches", {}).get("p50_ms"),
               "p50_text_to_topk_graph_ms": (text_lat or {}).get("hip_graph", {}).get("p50_ms"),
               "text_to_topk": text_lat, "roofline": roof, "cpu_baseline": cpu, "merged_vs_oracle": merged_check,
               "worst_case_uniform_codes": worst, "built_index": built, "built_index_1M": built_1m, "batch_sweep": batch_sweep,
               "exchange": (None if not gather else "two-phase (global threshold: 2 all-gathers per batch)" if two_phase
                            else "single (north_star: 1 all-gather of the per-shard top-k per batch)"),
               "fixed_batch_32": fixed32, "single_exchange": single_exchange, "two_phase_exchange": two_phase_leg, "index_build": index_build,
               "setup_seconds": {"generate": round(t_gen, 1), "upload_and_build": round(t_load, 1)},
               "hbm_bytes": device_bytes}
        sys.stdout.flush()
        os.dup2(real_stdout, 1)
        print(json.dumps(out), flush=True)
        os.dup2(2, 1)
    if s is not None:
        s.close()
    if gather:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
