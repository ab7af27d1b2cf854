"""Builds the 1 M-passage index on the device (indexer.index_device) and times the search of it at several batch sizes, one
batch at a time with per-kernel HIP-event times: does a batch whose fp16 score tables fit the Infinity Cache (8 MB per
query at K = 131 072; 256 MB of cache) run pass 1 faster per query?
    python tools/built_index_batch_probe.py [--docs 1000000] [--iters 4]"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--docs", type=int, default=1_000_000)
    ap.add_argument("--iters", type=int, default=4, help="k-means iterations (the search workload barely depends on them)")
    args = ap.parse_args()
    import torch

    import colbert_jl_amd as clb
    from colbert_jl_amd import indexer, synthetic
    from colbert_jl_amd.distributed import DeviceSearch
    dev = torch.device("cuda", 0)
    src = synthetic.DeviceMixtureSource(seed=61, n_docs=args.docs, device=dev)
    index, rec = indexer.index_device(src, nbits=2, kmeans_niters=args.iters, seed=62)
    s = clb.Searcher(index=index)
    host_small = {"centroids": np.asfortranarray(index["centroids"].cpu().numpy().T)}
    Q = synthetic.make_topic_queries(host_small["centroids"], seed=79, n_queries=256)
    Qdev = torch.from_numpy(np.ascontiguousarray(Q.transpose(2, 1, 0))).to(dev)
    out = {}
    for B in (4, 8, 16, 32, 64):
        ds = DeviceSearch(s, 32, B, 1000, 2)
        for i in range(3):
            ds(Qdev[i * B:(i + 1) * B])
        torch.cuda.synchronize()
        n = max(4, 128 // B)
        t0 = time.perf_counter()
        for i in range(n):
            ds(Qdev[(i * B) % (256 - B + 1):(i * B) % (256 - B + 1) + B])
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        s.profile_enable(True)
        for i in range(n):
            ds(Qdev[(i * B) % (256 - B + 1):(i * B) % (256 - B + 1) + B])
        torch.cuda.synchronize()
        prof = s.profile_read()
        s.profile_enable(False)
        out[B] = {"ms_per_batch": round(dt * 1e3, 3), "ms_per_query": round(dt * 1e3 / B, 4), "queries_per_s": round(B / dt, 1),
                  "kernels_ms_per_query": {k: round(v["ms"] / n / B, 4) for k, v in prof.items()}}
        print(B, json.dumps(out[B]), flush=True)
    s.close()


if __name__ == "__main__":
    main()
