#!/bin/bash
# Round 6: the epilogue of step i-1 under the MFMAs of step i (PIPE = 1: second accumulator) TOGETHER with the split table reads
# (-DCLB_APPROX_SPLIT_LUT: 16 VGPRs less, 164 in all -- no spills at three waves per SIMD), the combination VERDICT r05 asks for.
# Tuning build: make ABLATIONS=1 SUF=_abls EXTRA=-DCLB_APPROX_SPLIT_LUT.   -> gpurun_out/r06_pipe_split_ab.jsonl
set -u
OUT=gpurun_out/r06_pipe_split_ab.jsonl
: > $OUT
for W in headline uniform built100k; do
  case $W in
    headline) ARGS="";;
    uniform) ARGS="--uniform-codes";;
    built100k) ARGS="--built-docs 100000";;
  esac
  python3 tools/abl_sweep.py --tag "$W lib=product" $ARGS >> $OUT 2>> gpurun_out/r06_pipe_split_ab.err
  COLBERT_HIP_LIB=colbert.jl_amd/csrc/libcolbert_hip_abls.so python3 tools/abl_sweep.py --tag "$W lib=abl+split_lut" $ARGS --set CLB_DEBUG_APPROX_PIPE=0,1,0,1 >> $OUT 2>> gpurun_out/r06_pipe_split_ab.err
done
cat $OUT
