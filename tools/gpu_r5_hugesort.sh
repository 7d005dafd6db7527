#!/bin/bash
# round 5: in-memory class, scope of its fences: parity (every class entered), gene-window batch alone, the resident step
set -u
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_graph_m.py tests/test_full_scale.py -x -q -m gpu 2>&1 | tail -2
for fg in 1.0; do timeout 600 python tools/band_stats.py 262144 5000000 $fg 2>&1 | grep -E "later|stages"; done
timeout 900 python bench.py --steps 6 --warmup 3 --no-cpu-baseline --long-reads 0 --e2e-pairs 0 2>/dev/null | python3 -c '
import sys, json
j = json.loads(sys.stdin.read().strip().split("\n")[-1]); c = j["config"]
print("boundary", round(j["ms_per_step"], 1), "resident", round(c["resident"]["ms_per_step"], 1), "side", round(c["stage_ms"]["side_stream"], 1), "huge", round(c["stage_ms"]["dp_in_memory"], 1), "| gene", round(c.get("gene_window_pairs", {}).get("pairs_per_s", 0)), "backbone", round(c.get("backbone_pairs", {}).get("pairs_per_s", 0)), "ok", c["pairs_ok"], "errors", c["chain_errors"])'
