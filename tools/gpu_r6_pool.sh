#!/bin/bash
# round 6: the tail pool -- parity (pool sizes 2, 3, 4; the streamed sample; the whole alignment suite under HLALA_TAIL_POOL=3), then the resident step at k = 1, 2, 3, 4
set -u
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_graph_m.py -x -q -m gpu -k "tail_pool or streamed or two_batches" 2>&1 | tail -8 | tee gpurun_out/r6_pool_tests.log
HLALA_TAIL_POOL=3 timeout 1200 python -m pytest tests/test_gpu_align.py tests/test_graph_m.py -x -q -m gpu 2>&1 | tail -5 | tee -a gpurun_out/r6_pool_tests.log
for k in ${POOLS:-1 2 3 4}; do
  timeout 900 python bench.py --steps 12 --warmup 6 --no-extras --no-cpu-baseline --resident-only --tail-pool $k 2>gpurun_out/r6_pool_bench_$k.err | grep '^{' | tail -1 > gpurun_out/r6_pool_bench_$k.json
  python - $k <<'PY'
import json, sys
k = sys.argv[1]
try:
    j = json.load(open("gpurun_out/r6_pool_bench_%s.json" % k))
    print("tail pool %s: resident %.1f ms/step" % (k, j["ms_per_step"]), json.dumps({a: round(b, 1) for a, b in j["config"]["stage_ms"].items()}))
except Exception as e:
    print("tail pool", k, "failed:", e); print(open("gpurun_out/r6_pool_bench_%s.err" % k).read()[-2000:])
PY
done
