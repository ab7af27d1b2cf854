#!/bin/bash
# Round 6: the memory-path counters of pass 1 for both gather forms and both row formats on ONE workload (default: the headline).
# usage (on the GPU box): tools/r6_gather_pmc.sh [extra abl_sweep args]  -> gpurun_out/r06_gather_pmc.json
set -u
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r06_gather_pmc
rm -rf $OUT && mkdir -p $OUT
for CFG in "0 0" "1 0" "0 1" "1 1"; do
  set -- $CFG; GL=$1; ROWS=$2
  for C in "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES" "TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum TCP_TA_TCP_STATE_READ_sum TCP_GATE_EN1_sum"; do
    D=$OUT/gl${GL}_rows${ROWS}_$(echo $C | tr ' ' '_' | cut -c1-30)
    rocprofv3 --pmc $C --output-format csv -d $D -- python3 tools/abl_sweep.py --steps 6 --warm-seconds 0.02 --api pass1_gather=$GL --api score_rows=$ROWS "${@:3}" > $D.log 2> $D.err || echo "failed: $CFG $C" >> $OUT/failed.txt
  done
done
python3 - <<'PY'
import csv, glob, json, collections, os
out = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/r06_gather_pmc/**/*_counter_collection.csv", recursive=True):
    cfg = f.split("gpurun_out/r06_gather_pmc/")[1].split("/")[0].split("_")
    key = cfg[0] + "_" + cfg[1]
    for r in csv.DictReader(open(f)):
        want = "score_approx32_kernel<false, 0, %s, 0, %s>" % (cfg[0][2:], "true" if cfg[1] == "rows1" else "false")
        if want in r["Kernel_Name"]:
            out[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
            out[key]["dur_us"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
res = {k: {c: round(sum(v[-4:]) / max(len(v[-4:]), 1), 1) for c, v in d.items()} for k, d in out.items()}
for k in res:
    res[k]["launches_seen"] = len(out[k]["dur_us"])
json.dump(res, open("gpurun_out/r06_gather_pmc.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
find $OUT -name "*.csv" -size +2M -delete
