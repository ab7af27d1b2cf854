#!/bin/bash
# Round 6: pass-1 work-group size (12 / 8 / 4 waves within the registers of three waves per SIMD) x batches in flight (2 / 3):
# does a smaller pass-1 work-group let the other batches' kernels share the CUs?   -> gpurun_out/r06_bench_inflight.jsonl
set -u
OUT=gpurun_out/r06_bench_inflight.jsonl
: > $OUT
for NF in 2 3; do
for SUF in "" _w8 _w4 ""; do
  COLBERT_HIP_LIB=colbert.jl_amd/csrc/libcolbert_hip$SUF.so python3 bench.py --no-encoder --no-cpu --no-sub --no-latency --min-seconds 1.0 --in-flight $NF 2>> gpurun_out/r06_bench_inflight.err | \
    python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print(json.dumps({'lib':'${SUF:-product}','in_flight':$NF,'value':d['value'],'sustained':d['sustained']['value'],'one_batch_at_a_time':d['one_batch_at_a_time']['value'],'pass1_ms':d['roofline']['all_kernels_ms_per_step']['score_approx']}))" >> $OUT
done
done
cat $OUT
