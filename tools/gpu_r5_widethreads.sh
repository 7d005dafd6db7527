#!/bin/bash
# round 5: threads per DP call of the wide class (64 = one wavefront walks a frontier of up to 256 cells in four rounds): class time alone, resident step
set -u
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for x in ${WT_LIST:-128 256 64}; do
  touch hla-la_amd/csrc/kernel_dp.hip
  make -C hla-la_amd/csrc EXTRA="-DHLALA_DP_WIDE_THREADS=$x" 2>&1 | grep -E " error" | head
  echo "== build -DHLALA_DP_WIDE_THREADS=$x"
  timeout 600 python tools/band_stats.py 1048576 5000000 2>&1 | grep -E "later|stages|errors"
  timeout 900 python bench.py --steps 8 --warmup 3 --resident-only --no-cpu-baseline --long-reads 0 --no-extras 2>/dev/null | tail -1 > gpurun_out/r5_wt_$x.json
  python - <<PY
import json
d=json.load(open('gpurun_out/r5_wt_$x.json')); c=d['config']
print(" resident ms/step %.2f  stage_ms %s" % (d['ms_per_step'], {k: round(v, 1) for k, v in c['stage_ms'].items()}))
PY
done
