#!/usr/bin/env python3
"""Single-query latency of the search path on one GPU, with the per-kernel breakdown (HIP events inside the
library).  Usage: python tools/latency_breakdown.py [--docs N] [--batch B] [--reps R]"""
import argparse, json, os, sys, time
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--docs", type=int, default=1_000_000)
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--reps", type=int, default=60)
    ap.add_argument("--k", type=int, default=1000)
    ap.add_argument("--wide-select", type=int, default=-1, help="-1 by candidate capacity (default), 0 never, 1 always")
    args = ap.parse_args()
    import torch
    import colbert_jl_amd as clb
    from colbert_jl_amd import synthetic
    from colbert_jl_amd.distributed import DeviceSearch
    T, B = 32, args.batch
    idx = synthetic.make_index(seed=2024, n_docs=args.docs, n_blocks=8)
    s = clb.Searcher(index=idx, device=0)
    s.set_wide_select(args.wide_select)
    Q = synthetic.make_topic_queries(idx["centroids"], seed=77, n_queries=256, T=T)
    Qdev = torch.from_numpy(np.ascontiguousarray(Q.transpose(2, 1, 0))).cuda()
    run = DeviceSearch(s, T, B, args.k, 2)
    for i in range(5):
        run(Qdev[i * B:(i + 1) * B])
    torch.cuda.synchronize()
    lat = []
    for i in range(args.reps):
        q = Qdev[(i * B) % 200:(i * B) % 200 + B]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(q)
        torch.cuda.synchronize()
        lat.append(time.perf_counter() - t0)
    s.profile_enable(True)
    for i in range(args.reps):
        run(Qdev[(i * B) % 200:(i * B) % 200 + B])
        torch.cuda.synchronize()
    prof = s.profile_read()
    s.profile_enable(False)
    out = {"batch": B, "wide_select": args.wide_select, "p50_ms": round(float(np.median(lat)) * 1e3, 4), "min_ms": round(float(np.min(lat)) * 1e3, 4),
           "kernels_ms": {k: round(v["ms"] / max(v["launches"], 1), 4) for k, v in prof.items() if v["launches"]}}
    out["kernels_sum_ms"] = round(sum(out["kernels_ms"].values()), 4)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
