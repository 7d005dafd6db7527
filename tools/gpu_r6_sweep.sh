#!/bin/bash
# round 6: the parity sweep at full size on the final build (120 random worlds, none shared with the 48 under pytest), then 40 worlds each with the two-track band kernels switched on,
# with a tail pool of three, and with the agent-scope release build's switch settings left to tools/gpu_r6_agentrel.sh
set -u
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 2400 python tools/parity_sweep.py 120 600 20000 ) > gpurun_out/r6_parity_sweep.txt 2>&1
tail -1 gpurun_out/r6_parity_sweep.txt
( echo "== HLALA_DP_BAND2=1"; HLALA_DP_BAND2=1 timeout 1500 python tools/parity_sweep.py 40 1500 31000 ) >> gpurun_out/r6_parity_sweep.txt 2>&1
tail -1 gpurun_out/r6_parity_sweep.txt
( echo "== HLALA_TAIL_POOL=3 HLALA_DP_BAND2=1 HLALA_DP_BAND2_MARGIN=0"; HLALA_TAIL_POOL=3 HLALA_DP_BAND2=1 HLALA_DP_BAND2_MARGIN=0 timeout 1500 python tools/parity_sweep.py 30 1500 32000 ) >> gpurun_out/r6_parity_sweep.txt 2>&1
tail -1 gpurun_out/r6_parity_sweep.txt
