#!/usr/bin/env python3
"""Per-kernel time of ONE rank's work when the 1 M-passage corpus is sharded `--world` ways (the shard of rank 0 is
generated directly, as bench.py does): what a rank of an N-GPU run executes per 32-query batch, measured on one GPU.
Prints one JSON line."""
import argparse, json, os, sys, time
import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--docs", type=int, default=1_000_000)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--two-phase", action="store_true", help="cut at the global k-th approximate score (exchange precomputed)")
    ap.add_argument("--score-rows", type=int, default=-1, help="clb_searcher_set_score_rows: 0 fp16 rows, 1 8-bit cells (-1 default)")
    args = ap.parse_args()
    import torch
    import colbert_jl_amd as clb
    from colbert_jl_amd import synthetic
    from colbert_jl_amd.distributed import DeviceSearch
    T, B, k = 32, args.batch, 1000
    per = 8 // args.world
    K = synthetic.num_partitions_for(args.docs, 80.0)
    shard = synthetic.make_index(seed=2024, n_docs=args.docs, K=K, n_blocks=8, blocks=range(0, per))
    s = clb.Searcher(index=shard, device=0, pid_offset=int(shard["pid_offset"]))
    s.set_score_rows(args.score_rows)
    nb = 6                                                     # distinct batches
    Q = synthetic.make_topic_queries(shard["centroids"], seed=77, n_queries=max(256, nb * B), T=T)
    Qdev = torch.from_numpy(np.ascontiguousarray(Q.transpose(2, 1, 0))).cuda()
    run = DeviceSearch(s, T, B, k, 2)
    batches = [Qdev[i * B:(i + 1) * B] for i in range(nb)]
    tops = None
    if args.two_phase:
        # the exchange between the phases, precomputed: every shard's k largest approximate scores per query
        tops = [[] for _ in range(nb)]
        for rank in range(args.world):
            sh = shard if rank == 0 else synthetic.make_index(seed=2024, n_docs=args.docs, K=K, n_blocks=8,
                                                              blocks=range(rank * per, (rank + 1) * per))
            sr = s if rank == 0 else clb.Searcher(index=sh, device=0, pid_offset=int(sh["pid_offset"]))
            rr = run if rank == 0 else DeviceSearch(sr, T, B, k, 2)
            for i in range(nb):
                tops[i].append(rr.phase1(batches[i]).clone())
            torch.cuda.synchronize()
            if rank:
                sr.close()
        tops = [torch.stack(t) for t in tops]
        s.raise_bound_consts(s.bound_consts)      # one rank stands in for the group: marks the bound as shared

    def step(i):
        if args.two_phase:
            run.phase1(batches[i % nb])
            run.phase2(batches[i % nb], tops[i % nb])
        else:
            run(batches[i % nb])

    for i in range(3):
        step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    s.profile_enable(True, counters=True)
    for i in range(args.steps):
        step(i)
    torch.cuda.synchronize()
    prof = s.profile_read()
    stats = s.last_batch_stats()
    print(json.dumps({"world": args.world, "batch": B, "two_phase": bool(args.two_phase), "score_rows": s.score_rows, "shard_passages": int(shard["doclens"].size), "ms_per_batch": round(dt * 1e3, 4),
                      "queries_per_s_per_rank": round(B / dt, 1), "stats": stats,
                      "kernels_ms": {n: round(v["ms"] / max(v["launches"], 1), 4) for n, v in prof.items() if v["launches"]}}))


if __name__ == "__main__":
    main()
