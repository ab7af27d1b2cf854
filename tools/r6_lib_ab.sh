#!/bin/bash
# Round 6: A/B of experiment builds of the library (make SUF=_x EXTRA=...) against the product build, same box, same process order.
# usage (on the GPU box): tools/r6_lib_ab.sh <tag> <suffix> [<suffix> ...]   -> gpurun_out/r06_lib_ab_<tag>.jsonl
set -u
TAG=$1; shift
OUT=gpurun_out/r06_lib_ab_$TAG.jsonl
: > $OUT
for W in headline uniform built100k; do
  case $W in
    headline) ARGS="";;
    uniform) ARGS="--uniform-codes";;
    built100k) ARGS="--built-docs 100000";;
  esac
  for SUF in "" "$@" ""; do
    COLBERT_HIP_LIB=colbert.jl_amd/csrc/libcolbert_hip$SUF.so python3 tools/abl_sweep.py --tag "$W lib=${SUF:-product}" $ARGS >> $OUT 2>> gpurun_out/r06_lib_ab_$TAG.err
  done
done
cat $OUT
