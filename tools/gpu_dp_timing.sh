#!/bin/bash
set -u
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd $GRAFT_REPO_ROOT
# timing build of the library in a scratch copy (compile-time clocks in k_dp), then the normal build's numbers next to it
mkdir -p gpurun_out /tmp/tb && cp -r hla-la_amd include tools tests /tmp/tb/ && cd /tmp/tb
make -C hla-la_amd/csrc EXTRA=-DHLALA_DP_TIMING ../libhlala_gpu.so 2>&1 | tail -1
timeout 300 python tools/dbg_timing.py 262144 5000000 2>&1 | tail -6
