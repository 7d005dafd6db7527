#!/bin/bash
# projection kernel: time and cycles per level of the chunked form against the number of waves per CU (HLALA_PROJ_WAVES), gene-window pairs
set -u
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
for w in 14 10 7 4 2; do
  echo "-- HLALA_PROJ_WAVES=$w" | tee -a gpurun_out/r3_proj_waves.log
  ( HLALA_PROJ_WAVES=$w HLALA_DEBUG=1 timeout 900 python tools/dbg_timing.py 262144 5000000 m 1.0 2>&1 | grep -E "^ms |project" ) | tee -a gpurun_out/r3_proj_waves.log
done
