#!/bin/bash
# round 5: what the calls of the GENERAL 16-lane list cost on backbone pairs (profile build of tier 0, general instantiation only, every call recorded up to 500)
set -u
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd $GRAFT_REPO_ROOT
touch hla-la_amd/csrc/kernel_dp.hip
make -C hla-la_amd/csrc EXTRA="-DHLALA_DP_PROFILE=0 -DHLALA_DP_PROFILE_LOG2=4" 2>&1 | grep -E "error" | head
HLALA_DEBUG=1 timeout 600 python tools/dp_profile.py 262144 0.0 2>&1 | tail -12
