#!/bin/bash
# Round 6: the pass-1 ablation ladder on the round's final kernel (results staged in LDS, contiguous wave ranges), headline
# corpus and the k-means-built 1 M index.  Tuning build only (make ABLATIONS=1).
# usage (on the GPU box): tools/r6_pass1_study.sh   -> gpurun_out/r06_pass1_ablations.jsonl
set -u
export COLBERT_HIP_LIB=colbert.jl_amd/csrc/libcolbert_hip_abl.so
OUT=gpurun_out/r06_pass1_ablations.jsonl
: > $OUT
python3 tools/abl_sweep.py --tag headline --stats --set CLB_DEBUG_APPROX_VARIANT=0,1,3,5,7,8,10,0 >> $OUT 2> gpurun_out/r06_abl_headline.err
python3 tools/abl_sweep.py --tag built_index_1M --built-docs 1000000 --kmeans-iters 8 --stats --set CLB_DEBUG_APPROX_VARIANT=0,1,3,5,7,8,0 >> $OUT 2> gpurun_out/r06_abl_built.err
tail -3 $OUT
