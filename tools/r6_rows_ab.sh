#!/bin/bash
# Round 6: 8-bit (32-byte) score rows against fp16 (64-byte) rows, both gather forms, product build, four workloads.
# usage (on the GPU box): tools/r6_rows_ab.sh   -> gpurun_out/r06_score_rows_ab.jsonl
set -u
OUT=gpurun_out/r06_score_rows_ab.jsonl
: > $OUT
python3 tools/abl_sweep.py --tag headline --stats --api score_rows=0,1 --api pass1_gather=0,1 >> $OUT 2> gpurun_out/r06_rows_headline.err
python3 tools/abl_sweep.py --tag uniform_codes --uniform-codes --stats --api score_rows=0,1 --api pass1_gather=0,1 >> $OUT 2> gpurun_out/r06_rows_uniform.err
python3 tools/abl_sweep.py --tag built_index_100k --built-docs 100000 --stats --api score_rows=0,1 --api pass1_gather=0,1 >> $OUT 2> gpurun_out/r06_rows_built100k.err
python3 tools/abl_sweep.py --tag built_index_1M --built-docs 1000000 --kmeans-iters 8 --stats --api score_rows=0,1 --api pass1_gather=0,1 >> $OUT 2> gpurun_out/r06_rows_built1m.err
cat $OUT
