#!/bin/bash
# round 5: blocks of the wide class per CU (22 KB of LDS each; seven = 154 of a CU's 160 KB) against the LDS the main stream's kernels need beside it: resident step and stage times
set -u
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for v in ${WB_LIST:-7 6 5 4}; do
  echo "== HLALA_DP_WIDE_BLOCKS=$v ${SP_ENV:-}"
  env HLALA_DP_WIDE_BLOCKS=$v ${SP_ENV:-} timeout 900 python bench.py --steps 10 --warmup 3 --resident-only --no-cpu-baseline --long-reads 0 --no-extras 2>/dev/null | tail -1 > gpurun_out/r5_wb_$v.json
  python - <<PY
import json
d=json.load(open('gpurun_out/r5_wb_$v.json')); c=d['config']
print(" resident ms/step %.2f  stage_ms %s" % (d['ms_per_step'], {k: round(x, 1) for k, x in c['stage_ms'].items()}))
PY
done
