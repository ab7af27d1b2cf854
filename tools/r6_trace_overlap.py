#!/usr/bin/env python3
"""Reads a rocprofv3 --kernel-trace CSV of bench.py (two batches in flight) and reports, for the timed steady state, how much of
every kernel's run time overlapped another kernel's, and the union busy time per step.

    python tools/r6_trace_overlap.py <dir-with-*_kernel_trace.csv> [n_last_dispatches]
"""
import csv
import glob
import sys
from collections import defaultdict


def short(n):
    for k in ("score_approx32_kernel<true", "score_approx32_kernel<false", "score_exact_flat", "centroid_top_bf16x3_teams", "top_refine",
              "mark_count", "bitmap_emit", "select_margin", "topk_rank", "topk_kernel", "requantise", "token_range"):
        if k in n:
            return k
    return n.split("(")[0][-40:]


def main():
    files = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)
    rows = []
    for f in files:
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])))
    rows.sort()
    n_last = int(sys.argv[2]) if len(sys.argv) > 2 else 400
    rows = rows[-n_last:]
    t0, t1 = rows[0][0], max(r[1] for r in rows)
    # sweep: time with >= 1 and >= 2 kernels running
    ev = []
    for s, e, _ in rows:
        ev.append((s, 1)); ev.append((e, -1))
    ev.sort()
    busy1 = busy2 = 0
    depth, last = 0, ev[0][0]
    for t, d in ev:
        if depth >= 1: busy1 += t - last
        if depth >= 2: busy2 += t - last
        depth += d; last = t
    tot = defaultdict(int); ovl = defaultdict(int); cnt = defaultdict(int)
    for i, (s, e, n) in enumerate(rows):
        tot[n] += e - s; cnt[n] += 1
        for j, (s2, e2, n2) in enumerate(rows):
            if i != j and s2 < e and e2 > s:
                ovl[n] += min(e, e2) - max(s, s2)
    print(f"span {(t1 - t0) / 1e6:.3f} ms, some kernel running {busy1 / 1e6:.3f} ms, two or more {busy2 / 1e6:.3f} ms")
    for n in sorted(tot, key=lambda k: -tot[k]):
        print(f"{n:38s} n={cnt[n]:4d} avg {tot[n] / cnt[n] / 1e3:8.1f} us  overlapped with another kernel {100.0 * ovl[n] / max(tot[n], 1):5.1f} %")


if __name__ == "__main__":
    main()
