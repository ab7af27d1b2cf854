"""The general-shape search path (csrc/generic_kernels.hpp) on a dim-64 / nbits-8 index: queries/s and, under
rocprofv3 --kernel-trace --stats, where the time goes.  python tools/profile_generic_path.py [docs] [batch]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import colbert_jl_amd as clb  # noqa: E402

docs = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
idx = clb.synthetic.make_index(seed=5, n_docs=docs, dim=64, nbits=8)
Q = clb.synthetic.make_queries(idx, seed=6, n_queries=B)
s = clb.Searcher(index=idx, device=0)
s.search_batch(Q, 100, nprobe=2)
t0 = time.perf_counter()
reps = 3
for _ in range(reps):
    p, sc, nc = s.search_batch(Q, 100, nprobe=2)
dt = (time.perf_counter() - t0) / reps
print(f"generic path: {docs} passages, dim 64, nbits 8, K={idx['centroids'].shape[1]}: {dt / B * 1e3:.3f} ms per query ({B / dt:.1f} queries/s), "
      f"{nc.mean():.0f} candidates per query, checksum {int(p.sum())} {float(sc.sum()):.6f}")
s.close()
