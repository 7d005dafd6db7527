#!/bin/bash
# round 6: LDS-only synchronisation in the level loops of the projection kernels: parity, then the stage times of the resident loop
set -u
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1800 python -m pytest tests/test_gpu_align.py tests/test_graph_m.py tests/test_parity_sweep.py -x -q -m gpu 2>&1 | tail -5
timeout 900 python bench.py --steps 10 --warmup 4 --no-extras --no-cpu-baseline --resident-only 2>gpurun_out/r6_proj_bench.err | grep '^{' | tail -1 > gpurun_out/r6_proj_bench.json
python - <<'PY'
import json
j = json.load(open("gpurun_out/r6_proj_bench.json"))
print("resident %.1f ms/step" % j["ms_per_step"], json.dumps({a: round(b, 1) for a, b in j["config"]["stage_ms"].items()}))
PY
timeout 600 python tools/band_stats.py 1048576 5000000 2>&1 | grep -E "stages"
