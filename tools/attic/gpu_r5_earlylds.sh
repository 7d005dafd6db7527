#!/bin/bash
# round 5: the first early cells of a call mirrored in LDS (HLALA_DP_EARLY_LDS entries; 0 = the slab's hash only): parity, then class times of one batch alone and the resident step
set -u
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1800 python -m pytest tests/test_gpu_align.py tests/test_gpu_extend.py tests/test_graph_m.py tests/test_parity_sweep.py -m gpu -x -q > gpurun_out/r5_early_pytest.log 2>&1
tail -3 gpurun_out/r5_early_pytest.log
for x in ${EL_LIST:-16 0 8 32}; do
  touch hla-la_amd/csrc/kernel_dp.hip
  make -C hla-la_amd/csrc EXTRA="-DHLALA_DP_EARLY_LDS=$x" 2>&1 | grep -E " error" | head
  echo "== build -DHLALA_DP_EARLY_LDS=$x"
  timeout 600 python tools/band_stats.py 1048576 5000000 2>&1 | grep -E "16-lane|later|stages|errors"
  timeout 900 python bench.py --steps 8 --warmup 3 --resident-only --no-cpu-baseline --long-reads 0 --no-extras 2>/dev/null | tail -1 > gpurun_out/r5_el_$x.json
  python - <<PY
import json
d=json.load(open('gpurun_out/r5_el_$x.json')); c=d['config']
print(" resident ms/step %.2f  stage_ms %s" % (d['ms_per_step'], {k: round(v, 1) for k, v in c['stage_ms'].items()}))
PY
done
