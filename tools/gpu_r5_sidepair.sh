#!/bin/bash
# round 5: the side-stream classes queued after the main stream's stitch + pairing passes (HLALA_SIDE_AFTER_PAIR=1, default) or beside them (0): parity, resident step, stage times
set -u
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_align.py tests/test_graph_m.py -m gpu -x -q > gpurun_out/r5_sidepair_pytest.log 2>&1
tail -3 gpurun_out/r5_sidepair_pytest.log
for v in 0 1 0 1; do
  echo "== HLALA_SIDE_AFTER_PAIR=$v ${SP_ENV:-}"
  env HLALA_SIDE_AFTER_PAIR=$v ${SP_ENV:-} timeout 900 python bench.py --steps 10 --warmup 3 --resident-only --no-cpu-baseline --long-reads 0 --no-extras 2>/dev/null | tail -1 > gpurun_out/r5_sp_$v.json
  python - <<PY
import json
d=json.load(open('gpurun_out/r5_sp_$v.json')); c=d['config']
print(" resident ms/step %.2f  stage_ms %s" % (d['ms_per_step'], {k: round(x, 1) for k, x in c['stage_ms'].items()}))
PY
done
