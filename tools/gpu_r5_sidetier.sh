#!/bin/bash
# round 5: first class of the side stream (HLALA_DP_SIDE_TIER: 4 = broad, the default; 3 = wide as well): resident step, stage times, then the gene-window / backbone records
set -u
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd $GRAFT_REPO_ROOT
for x in "-DHLALA_DP_SIDE_TIER=2" "-DHLALA_DP_SIDE_TIER=3"; do
  touch hla-la_amd/csrc/kernel_dp.hip
  make -C hla-la_amd/csrc EXTRA="$x" 2>&1 | grep -E "error" | head
  echo "== EXTRA=$x"
  timeout 900 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --long-reads 0 --e2e-pairs 0 2>/dev/null | python3 -c '
import sys, json
j = json.loads(sys.stdin.read().strip().split("\n")[-1]); c = j["config"]
print("boundary", round(j["ms_per_step"], 1), "resident", round(c["resident"]["ms_per_step"], 1), "side", round(c["stage_ms"]["side_stream"], 1), "wide", round(c["stage_ms"]["dp_wide"], 1), "| gene", round(c.get("gene_window_pairs", {}).get("pairs_per_s", 0)), "backbone", round(c.get("backbone_pairs", {}).get("pairs_per_s", 0)))'
done
