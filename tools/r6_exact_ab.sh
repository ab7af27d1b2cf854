#!/bin/bash
# Round 6: the exact kernel with the rows of consecutive passages packed into one step (product build) against the build before
# (make SUF=_old from the parent commit), same box, A / B / A / B.   -> gpurun_out/r06_exact_packing_ab.jsonl
set -u
OUT=gpurun_out/r06_exact_packing_ab.jsonl
: > $OUT
for W in "" "--uniform-codes" "--built-docs 1000000"; do
for SUF in _old "" _old ""; do
  COLBERT_HIP_LIB=colbert.jl_amd/csrc/libcolbert_hip$SUF.so python3 tools/abl_sweep.py --api score_rows=0 --tag "${W:-headline}" --stats $W 2>> gpurun_out/r06_exact_packing_ab.err | \
    python3 -c "
import json,sys
for l in sys.stdin:
    l=l.strip()
    if not l.startswith('{') or 'score_exact' not in l: continue
    d=json.loads(l); d['lib']='${SUF:-packed}'; print(json.dumps({k:d[k] for k in ('lib','workload','score_exact','rescore_rows','score_approx','total','rescored_passages_per_query','rescored_rows_per_query') if k in d}))" >> $OUT
done
done
cat $OUT
