#!/bin/bash
# round 5 experiments (timing only, results wrong): the band kernel without its column stores / without the reads of the duplicate lists
set -u
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd $GRAFT_REPO_ROOT
for x in "" "-DBAND_X_NOCOLS" "-DBAND_X_NOALIAS" "-DBAND_X_NOCOLS -DBAND_X_NOALIAS"; do
  touch hla-la_amd/csrc/kernel_dp_band.hip
  make -C hla-la_amd/csrc EXTRA="$x" 2>&1 | grep -E "error" | head
  echo "== EXTRA=$x"
  timeout 300 python tools/band_stats.py 262144 5000000 2>&1 | grep -E "band:"
done
