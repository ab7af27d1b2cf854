#!/usr/bin/env python3
"""Summarises rocprofv3 output directories into one JSON for profiles/.

    python tools/pmc_summary.py <out.json> <dir> [<dir> ...]

Each <dir> is a `rocprofv3 -d` output (kernel-trace/stats CSVs and/or PMC counter_collection CSVs of
`bench.py --no-latency --no-cpu`, so every launch is a full batch).  Per kernel: launches, average
duration, and the average of every collected counter.  FETCH_SIZE / WRITE_SIZE are reported in KB by
rocprofv3; `hbm_read_bytes` applies the gfx950 correction for wide streaming reads (FETCH_SIZE counts
128-B requests as 64 B: x2, MI355X_MICROARCH.md section HBM) -- an upper estimate for gather-heavy kernels."""
import collections
import csv
import glob
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# rocprofv3 kernel name -> the kernel ids bench.py reports
def bench_id(name: str):
    base = name.split("<")[0]
    if base == "score_approx32_kernel":
        return "rescore_rows" if name.startswith("score_approx32_kernel<true") else "score_approx"
    return {"score_exact_flat_kernel": "score_exact", "score_exact_kernel": "score_exact",
            "centroid_top_bf16x3_mq_kernel": "centroid_scores",
            "centroid_top_bf16x3_teams_kernel": "centroid_scores"}.get(base)


def csrc_hash() -> str:
    """sha256 over the sources of the search kernels (search.hip and everything it includes): a PMC summary is only
    valid for the sources it was measured on.  The encoder / codec / communicator files do not enter."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "colbert.jl_amd", "csrc")
    for f in ("approx_kernels.hpp", "codec_kernels.hpp", "common.hpp", "generic_kernels.hpp", "search.hip", "search_kernels.hpp",
              "sort.hpp"):
        h.update(f.encode()); h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()


def short(name: str) -> str:
    n = name.split("(")[0]
    return n.replace("void ", "").replace("clb::", "").strip()


def main():
    out, dirs = sys.argv[1], sys.argv[2:]
    kern = collections.defaultdict(lambda: {"launches": 0, "dur_ns": 0.0, "counters": collections.defaultdict(list)})
    for d in dirs:
        for f in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
            seen = set()
            for r in csv.DictReader(open(f)):
                k = short(r["Kernel_Name"])
                kern[k]["counters"][r["Counter_Name"]].append(float(r["Counter_Value"]))
                key = (f, r["Dispatch_Id"])
                if key not in seen:
                    seen.add(key)
                    kern[k]["launches"] += 1
                    kern[k]["dur_ns"] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        for f in glob.glob(os.path.join(d, "**", "*_kernel_stats.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                k = short(r["Name"])
                kern[k].setdefault("trace", {"calls": int(r["Calls"]), "avg_ns": float(r["AverageNs"]),
                                              "pct": float(r["Percentage"])})
    res = {}
    for k, v in kern.items():
        e = {}
        if v["launches"]:
            e["pmc_launches"] = v["launches"]
            e["pmc_avg_us"] = round(v["dur_ns"] / v["launches"] / 1e3, 2)
        for c, vals in v["counters"].items():
            e[c] = round(sum(vals) / len(vals), 2)
        if "FETCH_SIZE" in e:
            e["hbm_read_bytes_uncorrected"] = int(e["FETCH_SIZE"] * 1024)
            e["hbm_read_bytes"] = int(e["FETCH_SIZE"] * 1024 * 2)
        if "WRITE_SIZE" in e:
            e["hbm_write_bytes"] = int(e["WRITE_SIZE"] * 1024)
        if "SQ_VALU_MFMA_BUSY_CYCLES" in e and e.get("GRBM_GUI_ACTIVE"):
            # cycles the matrix pipes were busy (summed over the 1 024 SIMDs) over the cycles the launch took (GRBM_GUI_ACTIVE is
            # summed over the 8 XCDs): the MFMA utilisation of the kernel at the clock the chip held during it
            e["mfma_pipe_busy_frac"] = round(e["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * e["GRBM_GUI_ACTIVE"] / 8.0), 4)
            e["effective_clock_GHz"] = round(e["GRBM_GUI_ACTIVE"] / 8.0 / (e["pmc_avg_us"] * 1e3), 3) if e.get("pmc_avg_us") else None
        if "trace" in v:
            e["trace"] = v["trace"]
        res[k] = e
    kernels = {}
    for k, e in res.items():
        bid = bench_id(k)
        if bid and ("FETCH_SIZE" in e or "WRITE_SIZE" in e):
            dst = kernels.setdefault(bid, {})
            if e.get("pmc_launches", 0) >= dst.get("pmc_launches", 0):     # the variant with the most launches
                dst.update(e, rocprof_name=k)
    doc = {"csrc_sha256": csrc_hash(),
           "correction": "calibrated with tools/microbench/fetch_calib.hip (profiles/r02_fetch_size_calibration.md): FETCH_SIZE "
                         "tallies every L2 miss request at 64 B -- contiguous streams (4- or 16-B-per-lane loads) leave L2 as "
                         "128-B requests and read 0.5x, 64-B row gathers read 1.0x.  hbm_read_bytes = FETCH_SIZE x 1024 x 2 is "
                         "the all-streaming figure; bench.py computes a mixed kernel's traffic as FETCH_SIZE x 1024 + half of "
                         "its known stream bytes; WRITE_SIZE x 1024",
           "kernels": kernels, "all": res}
    json.dump(doc, open(out, "w"), indent=1, sort_keys=True)
    print("wrote", out, len(res), "kernels")


if __name__ == "__main__":
    main()
