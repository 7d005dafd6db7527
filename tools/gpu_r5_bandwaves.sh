#!/bin/bash
# round 5: waves of the band kernels per CU, 1 M pairs
set -u
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd $GRAFT_REPO_ROOT
for w in 24 16 12 8; do
  echo "== HLALA_DP_BAND_WAVES=$w: $(HLALA_DP_BAND_WAVES=$w timeout 600 python tools/band_stats.py 1048576 5000000 2>&1 | grep -E 'band:')"
done
