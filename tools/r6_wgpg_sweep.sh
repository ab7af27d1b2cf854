#!/bin/bash
# Round 6: does pass 1 leaving a few CUs free (fewer than 32 persistent work-groups per XCD) let the OTHER batch's small,
# latency-bound kernels run beside it?  Tuning build (CLB_DEBUG_APPROX_WGPG), two batches in flight, same box, A/B/A.
#   -> gpurun_out/r06_wgpg_sweep.jsonl
set -u
OUT=gpurun_out/r06_wgpg_sweep.jsonl
: > $OUT
for W in 32 31 30 28 24 32; do
  CLB_DEBUG_APPROX_WGPG=$W COLBERT_HIP_LIB=colbert.jl_amd/csrc/libcolbert_hip_abl.so python3 bench.py --no-encoder --no-cpu --no-sub --no-latency --min-seconds 1.0 2>> gpurun_out/r06_wgpg_sweep.err | \
    python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print(json.dumps({'wgpg':$W,'value':d['value'],'ms_per_step':d['ms_per_step'],'sustained':d['sustained']['value'],'one_batch_at_a_time':d['one_batch_at_a_time']['value'],'pass1_ms':d['roofline']['all_kernels_ms_per_step']['score_approx']}))" >> $OUT
done
cat $OUT
