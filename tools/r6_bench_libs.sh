#!/bin/bash
# Round 6: the headline measurement (bench.py's search-only legs) with several builds of the library, same box, back to back.
# usage (on the GPU box): tools/r6_bench_libs.sh <tag> <suffix> [...]   -> gpurun_out/r06_bench_libs_<tag>.jsonl
set -u
TAG=$1; shift
OUT=gpurun_out/r06_bench_libs_$TAG.jsonl
: > $OUT
for SUF in "" "$@" ""; do
  COLBERT_HIP_LIB=colbert.jl_amd/csrc/libcolbert_hip$SUF.so python3 bench.py --no-encoder --no-cpu --no-sub --no-latency --min-seconds 1.0 2>> gpurun_out/r06_bench_libs_$TAG.err | \
    python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print(json.dumps({'lib':'${SUF:-product}','value':d['value'],'ms_per_step':d['ms_per_step'],'sustained':d['sustained']['value'],'one_batch_at_a_time':d['one_batch_at_a_time'],'kernels':d['roofline']['all_kernels_ms_per_step'],'in_flight_matches_serial':d['in_flight_matches_serial']}))" >> $OUT
done
cat $OUT
