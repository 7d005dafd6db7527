#!/bin/bash
# round 5: threads per DP call of the in-memory class (the slowest single call bounds a gene-window batch): resident step and the gene-window record
set -u
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd $GRAFT_REPO_ROOT
for x in "" "-DHLALA_DP_HUGE_THREADS=256" "-DHLALA_DP_HUGE_THREADS=512"; do
  touch hla-la_amd/csrc/kernel_dp.hip
  make -C hla-la_amd/csrc EXTRA="$x" 2>&1 | grep -E "error" | head
  echo "== EXTRA=$x"
  timeout 900 python bench.py --steps 6 --warmup 2 --resident-only --no-cpu-baseline --long-reads 0 --e2e-pairs 0 2>/dev/null | python3 -c '
import sys, json
j = json.loads(sys.stdin.read().strip().split("\n")[-1]); c = j["config"]
print("step", round(j["ms_per_step"], 1), "side", round(c["stage_ms"]["side_stream"], 1), "huge", round(c["stage_ms"]["dp_in_memory"], 1), "| gene", round(c.get("gene_window_pairs", {}).get("pairs_per_s", 0)), "backbone", round(c.get("backbone_pairs", {}).get("pairs_per_s", 0)))'
done
