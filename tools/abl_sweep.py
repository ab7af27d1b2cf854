#!/usr/bin/env python3
"""Sweeps the CLB_DEBUG_* knobs / ablation variants of the tuning build (make -C colbert.jl_amd/csrc ABLATIONS=1)
over bench.py's workload in ONE process: per setting, the average HIP-event time of every search kernel.

    COLBERT_HIP_LIB=colbert.jl_amd/csrc/libcolbert_hip_abl.so python tools/abl_sweep.py [--docs N] \
        --set CLB_DEBUG_APPROX_VARIANT=0,1,2 --set CLB_DEBUG_APPROX_WGPG=64,128

Results of ablation variants are wrong by design; nothing here is a product path."""
import argparse
import itertools
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--docs", type=int, default=1_000_000)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--set", action="append", default=[], help="NAME=v1,v2,...")
    ap.add_argument("--api", action="append", default=[],
                    help="score_rows=0,1 / pass1_gather=0,1 / centroid_products=1,3: product-library setters swept like --set "
                         "(results stay correct; works on the product build)")
    ap.add_argument("--uniform-codes", action="store_true")
    ap.add_argument("--built-docs", type=int, default=0,
                    help="instead of the generator-made index: this many passages of mixture embeddings through the repo's own "
                         "device-resident build (bench.py's built_index_1M workload at 1000000)")
    ap.add_argument("--kmeans-iters", type=int, default=20)
    ap.add_argument("--tag", default="", help="copied into every row (workload label)")
    ap.add_argument("--cell-range", action="store_true", help="print the half step of an 8-bit linear score table for this batch")
    ap.add_argument("--stats", action="store_true", help="add candidate / re-scored counts per query to every row")
    ap.add_argument("--warm-seconds", type=float, default=1.0, help="untimed batches before the first setting (profiling runs: keep it short)")
    args = ap.parse_args()
    import torch
    import colbert_jl_amd as clb
    from colbert_jl_amd import synthetic
    from colbert_jl_amd.distributed import DeviceSearch
    B, T, k = args.batch, 32, 1000
    if args.built_docs:
        from colbert_jl_amd import indexer
        dev = torch.device("cuda", 0)
        src = synthetic.DeviceMixtureSource(seed=61, n_docs=args.built_docs, device=dev)
        didx, rec = indexer.index_device(src, nbits=2, kmeans_niters=args.kmeans_iters, seed=62)
        s = clb.Searcher(index=didx)
        idx = indexer.index_to_host(didx)
        del didx
        Q = synthetic.make_queries(idx, seed=79, n_queries=B * 8, T=T)
        print(json.dumps({"built": {k_: rec[k_] for k_ in ("passages", "embeddings", "K", "kmeans_iters", "total_build_s")},
                          "pass1_gather_default": ["vgpr", "lds-dma"][s.pass1_gather[0]]}), flush=True)
    else:
        K = synthetic.num_partitions_for(args.docs, 80.0)
        idx = synthetic.make_index(seed=2024, n_docs=args.docs, K=K, n_blocks=8, topical=not args.uniform_codes)
        s = clb.Searcher(index=idx)
        Q = synthetic.make_topic_queries(idx["centroids"], seed=77, n_queries=B * 8, T=T)
    Qdev = torch.from_numpy(np.ascontiguousarray(Q.transpose(2, 1, 0))).cuda()
    if args.cell_range:
        # what an 8-bit linear score table (per-token range over the K centroids, 255 steps) would add to the per-cell bound
        Cd = torch.from_numpy(np.ascontiguousarray(np.asarray(idx["centroids"]).T)).cuda()          # (K, dim)
        half = []
        for b in range(B):
            cells = Qdev[b] @ Cd.T                                                                   # (T, K)
            half.append(((cells.max(dim=1).values - cells.min(dim=1).values) / 510.0).cpu().numpy())
        half = np.stack(half)
        print(json.dumps({"cell8_half_step": {"max_over_batch": float(half.max()), "mean": float(half.mean()),
                                              "note": "(max - min over the K centroids of Q_t.c) / 510, per (query, token)"}}), flush=True)
        del Cd
    run = DeviceSearch(s, T, B, k, 2)
    names = [x.split("=")[0] for x in args.set] + ["api:" + x.split("=")[0] for x in args.api]
    values = [x.split("=")[1].split(",") for x in args.set + args.api]
    # the first measurement of a process used to come out ~8 % slow (0.72 against 0.665 ms for the same pass 1: the device
    # has just spent seconds in host-side index generation and idles at a low clock): one second of untimed batches first
    import time
    t_end = time.time() + args.warm_seconds
    while time.time() < t_end:
        for i in range(8):
            run(Qdev[i * B:(i + 1) * B])
        torch.cuda.synchronize()
    for combo in itertools.product(*values) if values else [()]:
        for n, v in zip(names, combo):
            if n.startswith("api:"):
                getattr(s, "set_" + n[4:])(int(v))
            else:
                os.environ[n] = v
        for i in range(3):
            run(Qdev[i * B:(i + 1) * B])
        torch.cuda.synchronize()
        s.profile_enable(True)
        for i in range(args.steps):
            off = (i * B) % (Q.shape[2] - B + 1)
            run(Qdev[off:off + B])
        torch.cuda.synchronize()
        prof = s.profile_read()
        s.profile_enable(False)
        row = {n: v for n, v in zip(names, combo)}
        if args.tag:
            row["workload"] = args.tag
        row.update({kn: round(v["ms"] / max(v["launches"], 1), 4) for kn, v in prof.items() if v["launches"]})
        row["total"] = round(sum(v["ms"] for v in prof.values()) / args.steps, 4)
        if args.stats:
            s.profile_enable(True, counters=True)
            run(Qdev[0:B])
            torch.cuda.synchronize()
            st = s.last_batch_stats()
            s.profile_read(); s.profile_enable(False)
            row.update({"cand_embs_per_query": round(st["cand_embs"] / B), "rescored_passages_per_query": round(st["rescored_docs"] / B, 1),
                        "rescored_rows_per_query": round(st["rescored_embs"] / B, 1)})
        print(json.dumps(row), flush=True)
    s.close()


if __name__ == "__main__":
    main()
