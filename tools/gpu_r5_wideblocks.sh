#!/bin/bash
# round 5: blocks of the wide class per CU now that it runs on the side stream (7 = all the LDS it can get)
set -u
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd $GRAFT_REPO_ROOT
for w in 7 5 4 3; do
echo "== HLALA_DP_WIDE_BLOCKS=$w: $(HLALA_DP_WIDE_BLOCKS=$w timeout 900 python bench.py --steps 6 --warmup 3 --no-cpu-baseline --long-reads 0 --e2e-pairs 0 2>/dev/null | python3 -c '
import sys, json
j = json.loads(sys.stdin.read().strip().split("\n")[-1]); c = j["config"]
print("boundary", round(j["ms_per_step"], 1), "resident", round(c["resident"]["ms_per_step"], 1), "side", round(c["stage_ms"]["side_stream"], 1), "wide", round(c["stage_ms"]["dp_wide"], 1), "| gene", round(c.get("gene_window_pairs", {}).get("pairs_per_s", 0)), "backbone", round(c.get("backbone_pairs", {}).get("pairs_per_s", 0)))')"
done
