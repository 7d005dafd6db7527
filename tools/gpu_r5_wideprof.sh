#!/bin/bash
# round 5: where the calls of the wide class spend their cycles (profile build of tier 3, every call above 2^16 cycles, the first 500 recorded), mixed and gene-window pairs
set -u
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd $GRAFT_REPO_ROOT
touch hla-la_amd/csrc/kernel_dp.hip
make -C hla-la_amd/csrc EXTRA="-DHLALA_DP_PROFILE=${1:-3} -DHLALA_DP_PROFILE_LOG2=14" 2>&1 | grep -E " error" | head
for fg in 0.3 1.0; do
HLALA_DEBUG=1 timeout 600 python tools/dp_profile.py 262144 $fg 2>&1 | tail -${2:-28} | cut -c1-260
done
