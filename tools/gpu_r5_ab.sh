#!/bin/bash
# round 5: A/B of the band kernel on one box (HLALA_DP_BAND=0 / 1), one 262 k-pair Graph M batch alone, then 1 M pairs
set -u
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd $GRAFT_REPO_ROOT
for p in 262144 1048576; do
for v in 0 1; do
  echo "== pairs $p HLALA_DP_BAND=$v ${AB_ENV:-}"
  env HLALA_DP_BAND=$v ${AB_ENV:-} timeout 600 python tools/band_stats.py $p 5000000 2>&1 | grep -E "band:|16-lane|later|stages"
done
done
