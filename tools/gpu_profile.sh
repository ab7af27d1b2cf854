#!/bin/bash
# Profiles bench.py on the GPU box: kernel trace + stats, then PMC passes (each in its own run, as the pool requires).
# usage: tools/gpu_profile.sh <out-dir-under-gpurun_out> [extra bench.py args]
set -u
OUT=gpurun_out/$1; shift
mkdir -p $OUT
export TMPDIR=/tmp
BENCH="python3 bench.py --steps 5 --warmup 2 --no-cpu --no-latency --no-encoder --no-overlap --no-sub --min-seconds 0 $*"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $BENCH > $OUT/trace.json 2> $OUT/trace.err
# PMC_LIGHT=1: the two passes roofline.traffic needs (FETCH_SIZE, WRITE_SIZE + L2 hit / miss) -- the sub-records' workloads
if [ "${PMC_LIGHT:-0}" = "1" ]; then
  for C in "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"; do
    D=$OUT/pmc_$(echo $C | tr ' ' '_' | cut -c1-40)
    rocprofv3 --pmc $C --output-format csv -d $D -- $BENCH > $D.json 2> $D.err || echo "pmc pass failed: $C" >> $OUT/failed.txt
  done
  python3 tools/pmc_summary.py $OUT/pmc_summary.json $OUT/trace $OUT/pmc_* > $OUT/summary.log 2>&1
  find $OUT -name "*_agent_info.csv" -delete
  exit 0
fi
for C in "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_REQ_sum TCC_READ_sum" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TA_BUSY_avr" "GRBM_GUI_ACTIVE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES"; do
  D=$OUT/pmc_$(echo $C | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $C --output-format csv -d $D -- $BENCH > $D.json 2> $D.err || echo "pmc pass failed: $C" >> $OUT/failed.txt
done
python3 tools/pmc_summary.py $OUT/pmc_summary.json $OUT/trace $OUT/pmc_* > $OUT/summary.log 2>&1
# keep only the small per-kernel CSVs
find $OUT -name "*_agent_info.csv" -delete
