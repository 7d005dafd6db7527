#!/bin/bash
# broader random parity than the test suite: tools/stress_parity.py with the default classes and with the lane-per-DP class in front
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
( timeout 1200 python tools/stress_parity.py 6000 2>&1 | tail -10 ) | tee gpurun_out/r3_stress.log
( HLALA_DP_LANE=1 timeout 1200 python tools/stress_parity.py 4000 2>&1 | tail -10 ) | tee -a gpurun_out/r3_stress.log
