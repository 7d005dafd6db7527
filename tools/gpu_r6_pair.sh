#!/bin/bash
# round 6: the pairing stage split into k_pair_chains (one combination) + k_pair_multi (several): parity, then stage times of the resident loop
set -u
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_align.py tests/test_graph_m.py tests/test_unpaired.py tests/test_long_reads_full.py -x -q -m gpu 2>&1 | tail -8 | tee gpurun_out/r6_pair_tests.log
timeout 900 python bench.py --steps 8 --warmup 2 --no-extras --no-cpu-baseline --resident-only 2>gpurun_out/r6_pair_bench.err | grep '^{' | tail -1 > gpurun_out/r6_pair_bench.json
python - <<'PY'
import json
j = json.load(open("gpurun_out/r6_pair_bench.json"))
print("resident %.1f ms/step" % j["ms_per_step"])
print(json.dumps(j["config"]["stage_ms"]))
PY
timeout 600 python tools/band_stats.py 1048576 5000000 2>&1 | tail -8
