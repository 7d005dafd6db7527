#!/bin/bash
# round 5: column rows only for the chains that passed the filters (batch.h: chain_row) -- the whole GPU suite, then A/B against HLALA_ROWS_ALL=1
set -u
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r5_rows_pytest.log 2>&1
tail -8 gpurun_out/r5_rows_pytest.log
for v in 0 1; do
  echo "== HLALA_ROWS_ALL=$v"
  HLALA_ROWS_ALL=$v timeout 600 python tools/band_stats.py 1048576 5000000 2>&1 | grep -E "16-lane|later|stages"
  HLALA_ROWS_ALL=$v timeout 900 python bench.py --steps 10 --warmup 4 --no-extras --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r5_rows_bench_$v.json
  python - <<PY
import json
d=json.load(open('gpurun_out/r5_rows_bench_$v.json'))
print("value", d['value'], "ms", d['ms_per_step'], "resident", d.get('resident'), "stage_ms", d.get('stage_ms'))
PY
done
rocm-smi --showmeminfo vram 2>/dev/null | head -5
