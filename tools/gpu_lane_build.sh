#!/bin/bash
# the lane-per-DP class (kernel_dp_lane.hip) is left out of the default library; this builds a copy of the tree WITH it and runs its parity test
set -u
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd $GRAFT_REPO_ROOT
rm -rf /tmp/lane && mkdir -p /tmp/lane && cp -r hla-la_amd include tools tests oracle __graft_entry__.py /tmp/lane/
cd /tmp/lane && touch hla-la_amd/csrc/hlala_api.hip && make -C hla-la_amd/csrc EXTRA=-DHLALA_WITH_LANE_CLASS 2>&1 | grep -E "error|warning: v" ; make -C oracle 2>&1 | tail -1
timeout 900 python -m pytest tests/test_gpu_align.py -x -q -m gpu -k lane 2>&1 | tail -3
