#!/bin/bash
# round 5: chains per draw of k_project_chains (590 k draws per million pairs at 4: 39 per microsecond, one word hands out ~88)
set -u
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd $GRAFT_REPO_ROOT
for x in "-DHLALA_PROJ_DRAW=8" "-DHLALA_PROJ_DRAW=16" ""; do
  touch hla-la_amd/csrc/kernel_project.hip
  make -C hla-la_amd/csrc EXTRA="$x" 2>&1 | grep -E "error" | head
  echo "== EXTRA=$x: $(timeout 900 python bench.py --steps 6 --warmup 3 --resident-only --no-cpu-baseline --long-reads 0 --e2e-pairs 0 2>/dev/null | python3 -c '
import sys, json
j = json.loads(sys.stdin.read().strip().split("\n")[-1]); c = j["config"]
print("resident", round(j["ms_per_step"], 1), "project stage", round(c["stage_ms"]["project"], 1))')"
done
