#!/bin/bash
# round 6: parity tests after the housekeeping changes (lane class removed, debug entry points, pool trim across contexts), then a short resident + boundary bench
set -u
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_extend.py tests/test_gpu_align.py tests/test_graph_m.py tests/test_unpaired.py -x -q -m gpu 2>&1 | tail -15 | tee gpurun_out/r6_first_tests.log
timeout 900 python bench.py --steps 8 --warmup 2 --no-extras --no-cpu-baseline 2>gpurun_out/r6_first_bench.err | grep '^{' | tail -1 > gpurun_out/r6_first_bench.json
python - <<'PY'
import json
j = json.load(open("gpurun_out/r6_first_bench.json"))
print("boundary %.1f ms/step, resident %s" % (j["ms_per_step"], j["config"]["resident"]))
print(json.dumps(j["config"]["stage_ms"]))
PY
