"""The device-resident index build (indexer.index_device) at a given collection size, seconds per stage, optionally
followed by a search of the built index checked against the CPU oracle.
    python tools/bench_index_build_device.py --docs 1000000 --iters 20 --check 4 --out profiles/r04_index_build_1M.json"""
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
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--check", type=int, default=4, help="queries checked against the CPU oracle (0: none)")
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    import torch

    import colbert_jl_amd as clb
    from colbert_jl_amd import indexer, synthetic
    dev = torch.device("cuda", 0)
    t0 = time.time()
    src = synthetic.DeviceMixtureSource(seed=61, n_docs=args.docs, device=dev)
    index, rec = indexer.index_device(src, nbits=2, kmeans_niters=args.iters, seed=62, log=lambda m: print(m, flush=True))
    # the group lists of the nearest-centroid search: one fp16 product per fp32 product (nearest_top_f16_kernel, round 5) unless
    # COLBERT_NEAREST_PRODUCTS=3 asks for the three-product bf16 split; both run at the same dense MFMA peak
    products = 3 if os.environ.get("COLBERT_NEAREST_PRODUCTS") == "3" else 1
    what = "bf16, 3 products per fp32 product" if products == 3 else "fp16, 1 product per fp32 product"
    rec["nearest_products"] = products
    tf = products * 2.0 * 128 * rec["sample_points"] * rec["K"] * rec["kmeans_iters"] / rec["kmeans_s"] / 1e12
    rec["kmeans_roofline"] = {"bound": "mfma", "achieved": round(tf, 1), "peak": 2500.0,
                              "unit": f"TFLOP/s ({what}; the centroid update inside the time)", "frac": round(tf / 2500.0, 4)}
    ctf = products * 2.0 * 128 * rec["embeddings"] * rec["K"] / rec["compress_s"] / 1e12
    rec["compress_roofline"] = {"bound": "mfma", "achieved": round(ctf, 1), "peak": 2500.0, "unit": f"TFLOP/s ({what})",
                                "frac": round(ctf / 2500.0, 4)}
    t1 = time.time()
    s = clb.Searcher(index=index)
    torch.cuda.synchronize()
    rec["searcher_create_s"] = round(time.time() - t1, 3)
    rec["searcher_device_GB"] = round(s.device_bytes / 1e9, 2)
    rec["pass1_gather"] = s.pass1_gather
    if args.check:
        from oracle import oracle as orc
        orc.build()
        host = indexer.index_to_host(index)
        Q = synthetic.make_queries(host, seed=63, n_queries=args.check)
        oidx = dict(host, emb2pid=orc.build_emb2pid(host["doclens"]))
        ok = True
        for j in range(args.check):
            rp, rs, _ = orc.search(oidx, Q[:, :, j], 2, 1000)
            p, sc = s.search_embeddings(Q[:, :, j], k=1000)
            ok = ok and bool(np.array_equal(p, rp)) and bool(np.array_equal(sc.view(np.uint32), rs.view(np.uint32)))
        rec["search_matches_oracle"] = ok
        rec["queries_checked"] = args.check
    rec["wall_s"] = round(time.time() - t0, 1)
    print(json.dumps(rec))
    if args.out:
        with open(args.out, "w") as f:
            json.dump(rec, f, indent=1)
    s.close()


if __name__ == "__main__":
    main()
