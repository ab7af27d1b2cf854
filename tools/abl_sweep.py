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
    ap.add_argument("--uniform-codes", action="store_true")
    args = ap.parse_args()
    import torch
    import colbert_jl_amd as clb
    from colbert_jl_amd import synthetic
    from colbert_jl_amd.distributed import DeviceSearch
    K = synthetic.num_partitions_for(args.docs, 80.0)
    idx = synthetic.make_index(seed=2024, n_docs=args.docs, K=K, n_blocks=8, topical=not args.uniform_codes)
    s = clb.Searcher(index=idx)
    B, T, k = args.batch, 32, 1000
    Q = synthetic.make_topic_queries(idx["centroids"], seed=77, n_queries=B * 8, T=T)
    Qdev = torch.from_numpy(np.ascontiguousarray(Q.transpose(2, 1, 0))).cuda()
    run = DeviceSearch(s, T, B, k, 2)
    names = [x.split("=")[0] for x in args.set]
    values = [x.split("=")[1].split(",") for x in args.set]
    for combo in itertools.product(*values) if values else [()]:
        for n, v in zip(names, combo):
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
        row.update({kn: round(v["ms"] / max(v["launches"], 1), 4) for kn, v in prof.items() if v["launches"]})
        row["total"] = round(sum(v["ms"] for v in prof.values()) / args.steps, 4)
        print(json.dumps(row), flush=True)
    s.close()


if __name__ == "__main__":
    main()
