#!/usr/bin/env python3
"""Times the index-build stages (I1-I6 of SURVEY.md 8a) on the device at BASELINE config-2 scale:
synthetic 100k passages -> sample -> k-means -> codec statistics -> compress -> IVF.  Host<->device copies of
the stand-alone entry points are included (they take host buffers).  Prints one JSON line."""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import colbert_jl_amd as clb  # noqa: E402
from colbert_jl_amd import codec, synthetic  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--docs", type=int, default=100_000)
    ap.add_argument("--iters", type=int, default=4)
    args = ap.parse_args()
    t0 = time.time()
    embs, doclens = synthetic.make_embeddings(seed=61, n_docs=args.docs)
    gen = time.time() - t0
    n_emb = embs.shape[1]
    rng = np.random.default_rng(62)
    n_s = codec.num_sampled_pids(args.docs)
    off = np.concatenate([[0], np.cumsum(doclens)])
    pids = np.unique(rng.integers(0, args.docs, size=n_s))
    cols = np.concatenate([np.arange(off[p], off[p + 1]) for p in pids])
    sample = np.asfortranarray(embs[:, rng.permutation(cols)])
    h = codec.heldout_size(sample.shape[1])
    sample, held = sample[:, :-h], sample[:, -h:]
    plan = codec.setup(args.docs, float(doclens[pids].mean()), sample.shape[1], 25000, 1)
    K = plan["num_partitions"]
    init = sample[:, rng.permutation(sample.shape[1])[:K]]
    out = {"docs": args.docs, "embeddings": int(n_emb), "sample_points": int(sample.shape[1]), "K": int(K),
           "generate_s": round(gen, 1)}
    t0 = time.time(); cent, _, it = codec.kmeans(sample, init, max_iters=args.iters); dt = time.time() - t0
    out["kmeans_s_per_iter"] = round(dt / max(it, 1), 3); out["kmeans_iters"] = int(it)
    out["kmeans_tflops"] = round(2.0 * 128 * sample.shape[1] * K * it / dt / 1e12, 1)
    t0 = time.time(); cut, w, avg, _ = codec.compute_avg_residuals(2, cent, held); out["codec_stats_s"] = round(time.time() - t0, 3)
    t0 = time.time()
    chunk = 2_000_000
    codes = np.concatenate([codec.compress(cent, cut, 128, 2, embs[:, i:i + chunk])[0] for i in range(0, n_emb, chunk)])
    dt = time.time() - t0
    out["compress_s"] = round(dt, 3); out["compress_Memb_per_s"] = round(n_emb / dt / 1e6, 2)
    t0 = time.time(); ivf, lens = codec.build_ivf(codes, K); out["build_ivf_s"] = round(time.time() - t0, 3)
    assert lens.sum() == n_emb
    print(json.dumps(out))


if __name__ == "__main__":
    main()
