#!/bin/bash
# round 6: the long-read projection compiled for five wavefronts per SIMD (96 registers, 16 spilled; gpurun_in_ab/wps5.so) at 18 / 20 blocks per CU against the default build (four, 16 blocks)
set -u
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd $GRAFT_REPO_ROOT
for i in 1 2; do
  echo "default"; timeout 900 python tools/long_phase.py 50000 5000000 2>&1 | grep "reads/s"
  for w in 18 20 16; do echo "wps5, $w blocks per CU"; HLALA_LIB_PATH=$GRAFT_REPO_ROOT/gpurun_in_ab/wps5.so HLALA_PROJ_LONG_WAVES=$w timeout 900 python tools/long_phase.py 50000 5000000 2>&1 | grep "reads/s"; done
done
