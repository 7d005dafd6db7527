#!/bin/bash
# round 5: the boundary loop with two and with three alignments in flight; then the eight-sample test
set -u
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd $GRAFT_REPO_ROOT
for f in 2 3; do
  echo "== --in-flight $f"
  timeout 900 python bench.py --in-flight $f --steps 8 --warmup 3 --no-cpu-baseline --no-extras --long-reads 0 2>gpurun_out/r5_inflight_$f.err | python3 -c '
import sys, json
j = json.loads(sys.stdin.read().strip().split("\n")[-1])
print(round(j["value"]), round(j["ms_per_step"], 1), "resident", round(j["config"]["resident"]["ms_per_step"], 1), j["host_inclusive"]["host_thread_ms_per_call"])'
  tail -2 gpurun_out/r5_inflight_$f.err
done

