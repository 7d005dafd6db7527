#!/bin/bash
# round 5: reach margins of the jump-free list (a JF call that meets a jump now goes to the general 16-lane list, not to the 32-lane class) and of the band lists
set -u
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd $GRAFT_REPO_ROOT
for m in 16 8 4 1; do
  echo "== HLALA_DP_JF_MARGIN=$m"
  HLALA_DP_JF_MARGIN=$m timeout 600 python tools/band_stats.py 1048576 5000000 2>&1 | grep -E "16-lane|later|stages"
done
for m in 4 12; do
  echo "== HLALA_DP_BAND_MARGIN=$m"
  HLALA_DP_BAND_MARGIN=$m timeout 600 python tools/band_stats.py 1048576 5000000 2>&1 | grep -E "band:|16-lane|stages"
done
