#!/bin/bash
# round 5: 24-byte cell records in the 16- and 32-lane classes: parity, then class times alone and the resident step
set -u
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd $GRAFT_REPO_ROOT
timeout 1800 python -m pytest tests/test_gpu_extend.py tests/test_gpu_align.py tests/test_graph_m.py tests/test_parity_sweep.py -x -q -m gpu 2>&1 | tail -3
timeout 600 python tools/band_stats.py 1048576 5000000 2>&1 | grep -E "16-lane|later|stages"
timeout 900 python bench.py --steps 6 --warmup 3 --resident-only --no-cpu-baseline --long-reads 0 --e2e-pairs 0 2>/dev/null | python3 -c '
import sys, json
j = json.loads(sys.stdin.read().strip().split("\n")[-1]); c = j["config"]
print("resident", round(j["ms_per_step"], 1), "general", round(c["stage_ms"]["dp_16lane_general_part"], 1), "jf", round(c["stage_ms"]["dp_16lane_jump_free_part"], 1), "mid", round(c["stage_ms"]["dp_32lane"], 1))'
