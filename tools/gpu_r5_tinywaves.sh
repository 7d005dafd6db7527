#!/bin/bash
# round 5: waves per SIMD the GENERAL instantiation of the 16-lane class is compiled for, now that it holds 0.58 M long calls (3: 168 registers, nothing spilled; 4: 128, 10 spilled)
set -u
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd $GRAFT_REPO_ROOT
for x in "-DHLALA_DP_TINY_WAVES=3" ""; do
  touch hla-la_amd/csrc/kernel_dp.hip
  make -C hla-la_amd/csrc EXTRA="$x" 2>&1 | grep -E "error" | head
  echo "== EXTRA=$x: $(timeout 900 python bench.py --steps 6 --warmup 3 --resident-only --no-cpu-baseline --long-reads 0 --e2e-pairs 0 2>/dev/null | python3 -c '
import sys, json
j = json.loads(sys.stdin.read().strip().split("\n")[-1]); c = j["config"]
print("resident", round(j["ms_per_step"], 1), "general", round(c["stage_ms"]["dp_16lane_general_part"], 1), "jf", round(c["stage_ms"]["dp_16lane_jump_free_part"], 1))')"
done
