#!/bin/bash
# round 6: which calls the two-track band kernels should take: off / up to 15 / 31 / 63 read bases left; margins
set -u
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd $GRAFT_REPO_ROOT
for v in "HLALA_DP_BAND2=0" "HLALA_DP_BAND2_MAXJ=15" "HLALA_DP_BAND2_MAXJ=31" "HLALA_DP_BAND2_MAXJ=63" "HLALA_DP_BAND2_MAXJ=15 HLALA_DP_BAND2_MARGIN=24" "HLALA_DP_BAND2_MAXJ=15 HLALA_DP_BAND2_MARGIN=40"; do
  echo "== $v"
  env $v timeout 600 python tools/band_stats.py 1048576 5000000 2>&1 | grep -E "two-track band:|16-lane|later|stages"
done
