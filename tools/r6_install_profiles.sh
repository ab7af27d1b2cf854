#!/bin/bash
# Copies the summaries of a profile collection (tools/gpu_profile.sh <prefix>, <prefix>_uniform, <prefix>_built, <prefix>_built1m under
# gpurun_out/) into profiles/ under the names bench.py and profiles/README.md use.   usage: tools/r6_install_profiles.sh r06c_prof
set -eu
P=gpurun_out/$1
cp $P/pmc_summary.json profiles/pmc_summary.json
cp $P/pmc_summary.json profiles/r06_pmc_summary.json
cp $P/trace/runc/*_kernel_stats.csv profiles/r06_kernel_stats.csv
for p in "uniform:uniform_codes" "built:built_index" "built1m:built_index_1M"; do
  a=${p%%:*}; b=${p##*:}
  cp ${P}_$a/pmc_summary.json profiles/pmc_summary_$b.json
  cp ${P}_$a/pmc_summary.json profiles/r06_pmc_summary_$b.json
  cp ${P}_$a/trace/runc/*_kernel_stats.csv profiles/r06_kernel_stats_$b.csv
done
python3 - <<'PY'
import json, sys
sys.path.insert(0, '.')
from tools.pmc_summary import csrc_hash
h = csrc_hash()
for f in ("pmc_summary", "pmc_summary_uniform_codes", "pmc_summary_built_index", "pmc_summary_built_index_1M"):
    print(f, json.load(open(f"profiles/{f}.json")).get("csrc_sha256") == h)
PY
