#!/bin/bash
# Round 5: the pass-1 ablation ladder on the current kernel, headline corpus and the k-means-built 1 M index, plus the two
# trials VERDICT r04 asks for (8-bit / 32-byte score rows; row mask folded into pass 1).  Tuning build only.
# usage (on the GPU box): tools/r5_pass1_study.sh   -> gpurun_out/r05_pass1_ablations.jsonl
set -u
export COLBERT_HIP_LIB=colbert.jl_amd/csrc/libcolbert_hip_abl.so
OUT=gpurun_out/r05_pass1_ablations.jsonl
: > $OUT
python3 tools/abl_sweep.py --tag headline --stats --cell-range --set CLB_DEBUG_APPROX_VARIANT=0,1,3,5,7,8,9,10 >> $OUT 2> gpurun_out/r05_abl_headline.err
python3 tools/abl_sweep.py --tag headline_eps --stats --set CLB_DEBUG_EPS_T_ADD_1E6=0,1000,2000,3000,4000 >> $OUT 2>> gpurun_out/r05_abl_headline.err
python3 tools/abl_sweep.py --tag built_index_1M --built-docs 1000000 --kmeans-iters 8 --stats --cell-range --set CLB_DEBUG_APPROX_VARIANT=0,1,3,5,7,8,9,10 >> $OUT 2> gpurun_out/r05_abl_built.err
tail -3 $OUT
