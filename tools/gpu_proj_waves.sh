#!/bin/bash
# k_rethread_chains: time of the projection stage against its waves per CU (HLALA_RETHREAD_WAVES), gene-window pairs and the mixed workload
set -u
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
for w in 24 20 16 12 8; do
  echo "-- HLALA_RETHREAD_WAVES=$w" | tee -a gpurun_out/r3_proj_waves.log
  for cfg in "262144 5000000 m 1.0" "1048576 5000000 m 0.3"; do
  ( HLALA_RETHREAD_WAVES=$w timeout 900 python tools/dbg_timing.py $cfg 2>&1 | grep -E "^ms " ) | tee -a gpurun_out/r3_proj_waves.log
  done
done
