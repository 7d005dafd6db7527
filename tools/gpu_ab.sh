#!/bin/bash
# A/B of two versions of kernel_dp.hip on the same box: A = gpurun_in_kernel_dp_A.hip (repo root), B = the tree's file
set -u
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd $GRAFT_REPO_ROOT
N=${1:-1048576}
mkdir -p /tmp/a && cp -r hla-la_amd include tools tests /tmp/a/ && cp gpurun_in_kernel_dp_A.hip /tmp/a/hla-la_amd/csrc/kernel_dp.hip
[ -f gpurun_in_api_A.hip ] && cp gpurun_in_api_A.hip /tmp/a/hla-la_amd/csrc/hlala_api.hip
[ -f gpurun_in_batch_A.h ] && cp gpurun_in_batch_A.h /tmp/a/hla-la_amd/csrc/batch.h
( cd /tmp/a && touch hla-la_amd/csrc/kernel_dp.hip && make -C hla-la_amd/csrc ../libhlala_gpu.so 2>&1 | grep -E "error" )
for i in 1 2; do
  echo "A:"; ( cd /tmp/a && timeout 600 python tools/dbg_timing.py $N 5000000 2>&1 | tail -2 )
  echo "B:"; timeout 600 python tools/dbg_timing.py $N 5000000 2>&1 | tail -2
done
