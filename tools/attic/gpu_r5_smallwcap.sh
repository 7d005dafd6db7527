#!/bin/bash
# round 5: frontier capacity of the 64-lane class, 64 (one round per phase) / 128 (two rounds when a frontier needs them; twice the LDS per block): how many calls still reach the wide class,
# class times of one batch alone, resident step -- after the parity tests of the 128 build
set -u
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for x in ${SW_LIST:-128 64}; do
  touch hla-la_amd/csrc/kernel_dp.hip
  make -C hla-la_amd/csrc EXTRA="-DHLALA_DP_SMALL_WCAP=$x" 2>&1 | grep -E " error" | head
  echo "== build -DHLALA_DP_SMALL_WCAP=$x"
  if [ "$x" != "64" ]; then timeout 1200 python -m pytest tests/test_gpu_align.py tests/test_graph_m.py -m gpu -x -q 2>&1 | tail -2; fi
  timeout 600 python tools/band_stats.py 1048576 5000000 2>&1 | grep -E "later|stages|errors"
  timeout 600 python tools/band_stats.py 262144 5000000 1.0 2>&1 | grep -E "later|stages|errors"
  timeout 900 python bench.py --steps 8 --warmup 3 --resident-only --no-cpu-baseline --long-reads 0 --no-extras 2>/dev/null | tail -1 > gpurun_out/r5_sw_$x.json
  python - <<PY
import json
d=json.load(open('gpurun_out/r5_sw_$x.json')); c=d['config']
print(" resident ms/step %.2f  stage_ms %s" % (d['ms_per_step'], {k: round(v, 1) for k, v in c['stage_ms'].items()}))
PY
done
