#!/usr/bin/env python3
"""Latency of the search path as stream launches against a captured HIP graph (DeviceSearch.capture), batch size as the
first argument.  "graph replay only" re-runs the SAME query (warm caches): compare the line that copies a new query in."""
import sys, time, json, numpy as np
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import colbert_jl_amd as clb
from colbert_jl_amd import synthetic
from colbert_jl_amd.distributed import DeviceSearch
T, B, k = 32, int(sys.argv[1]) if len(sys.argv) > 1 else 1, 1000
idx = synthetic.make_index(seed=2024, n_docs=1_000_000, n_blocks=8)
s = clb.Searcher(index=idx, device=0)
Q = synthetic.make_topic_queries(idx["centroids"], seed=77, n_queries=256, T=T)
Qdev = torch.from_numpy(np.ascontiguousarray(Q.transpose(2, 1, 0))).cuda()
run = DeviceSearch(s, T, B, k, 2)
for i in range(5): run(Qdev[i * B:(i + 1) * B])
torch.cuda.synchronize()
def p50(fn, reps=80):
    lat = []
    for i in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(i); torch.cuda.synchronize(); lat.append(time.perf_counter() - t0)
    return round(float(np.median(lat)) * 1e3, 4), round(float(np.min(lat)) * 1e3, 4)
print("stream launches", p50(lambda i: run(Qdev[(i * B) % 200:(i * B) % 200 + B])))
static_q = Qdev[0:B].clone()
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
try:
    with torch.cuda.stream(side):
        run(static_q)
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        run(static_q)
    torch.cuda.synchronize()
    ref_p, ref_s = run(Qdev[B:2 * B]); ref_p = ref_p.clone(); ref_s = ref_s.clone()
    static_q.copy_(Qdev[B:2 * B]); g.replay(); torch.cuda.synchronize()
    print("graph result equal:", bool(torch.equal(run.out_p, ref_p) and torch.equal(run.out_s, ref_s)))
    def rep(i):
        static_q.copy_(Qdev[(i * B) % 200:(i * B) % 200 + B]); g.replay()
    print("graph replay (incl. query copy)", p50(rep))
    print("graph replay only", p50(lambda i: g.replay()))
except Exception as e:
    print("capture failed:", repr(e)[:400])
