#!/bin/bash
# round 5: band lists by the exact reach bound (default) against "as soon as the linear run covers the read bases" (HLALA_DP_BAND_RISKY=1: more calls, some fail over)
set -u
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd $GRAFT_REPO_ROOT
for v in 0 1; do
  echo "== HLALA_DP_BAND_RISKY=$v"
  HLALA_DP_BAND_RISKY=$v timeout 600 python tools/band_stats.py 1048576 5000000 2>&1 | grep -E "band:|fail-over|16-lane|later|stages"
done
