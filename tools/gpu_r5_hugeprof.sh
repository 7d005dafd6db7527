#!/bin/bash
# round 5: the slowest calls of the in-memory class (profile build of tier 6) on gene-window pairs
set -u
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd $GRAFT_REPO_ROOT
touch hla-la_amd/csrc/kernel_dp.hip
make -C hla-la_amd/csrc EXTRA="-DHLALA_DP_PROFILE=${1:-6}" 2>&1 | grep -E "error" | head
HLALA_DEBUG=1 timeout 600 python tools/dp_profile.py 262144 1.0 2>&1 | tail -45 | cut -c1-200
