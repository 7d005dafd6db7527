#!/bin/bash
# round 5: the runtime's copy kernels beside the persistent kernels: workgroups per copy kernel (DEBUG_CLR_LIMIT_BLIT_WG), boundary loop without a profiler
set -u
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd $GRAFT_REPO_ROOT
for v in "X=0" "DEBUG_CLR_LIMIT_BLIT_WG=4" "DEBUG_CLR_LIMIT_BLIT_WG=8" "DEBUG_CLR_LIMIT_BLIT_WG=16" "X=1"; do
  echo "== $v: $(env $v timeout 600 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-extras --long-reads 0 2>/dev/null | grep -h '^{' | tail -1 | python3 -c 'import sys,json; j=json.loads(sys.stdin.read()); print(round(j["value"]), round(j["ms_per_step"],1), "resident", round(j["config"]["resident"]["ms_per_step"],1), {k: round(v, 1) for k, v in j["host_inclusive"]["host_thread_ms_per_call"].items()})' 2>&1 | tail -1)"
done
