#!/bin/bash
# the 16-lane DP kernel against the number of its waves per CU (HLALA_TINY_WAVES_PER_CU; 16 = the default): is its time the latency of one wave (time ~ 1 / waves) or a shared resource?
set -u
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
for w in 16 12 8 4; do
  echo "-- HLALA_TINY_WAVES_PER_CU=$w" | tee -a gpurun_out/r3_tiny_waves.log
  ( HLALA_TINY_WAVES_PER_CU=$w timeout 900 python tools/dbg_timing.py 1048576 5000000 m 0.3 2>&1 | grep -E "^ms |^retry" ) | tee -a gpurun_out/r3_tiny_waves.log
done
