#!/bin/bash
# round 5: phase clocks of the band kernel (timing build) at several grid sizes
set -u
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd $GRAFT_REPO_ROOT
touch hla-la_amd/csrc/kernel_dp_band.hip
make -C hla-la_amd/csrc EXTRA="-DHLALA_BAND_TIMING ${BAND_EXTRA:-}" 2>&1 | grep -E "error" | head
for w in ${BAND_WAVES_LIST:-24 12 4}; do
  echo "== HLALA_DP_BAND_WAVES=$w"
  HLALA_DP_BAND_WAVES=$w timeout 300 python tools/band_stats.py 262144 5000000 2>&1 | grep -E "band:|band kernel"
done
