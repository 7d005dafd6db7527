#!/bin/bash
# phase clocks of the DP kernels (build with -DHLALA_DP_TIMING in a scratch copy): gpu_phase_clocks.sh <pairs> <levels> [m [frac_gene]]
set -u
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
make -s -C tools/graphm 2>&1 | tail -1
rm -rf /tmp/vt && mkdir /tmp/vt && cp -r hla-la_amd include tools tests __graft_entry__.py /tmp/vt/
( cd /tmp/vt && touch hla-la_amd/csrc/hlala_api.hip && make -s -C hla-la_amd/csrc ../libhlala_gpu.so EXTRA="-DHLALA_DP_TIMING" 2>&1 | grep -E "rror" )
( cd /tmp/vt && HLALA_DEBUG=1 timeout 900 python tools/dbg_timing.py "$@" 2>&1 | tail -8 ) | tee gpurun_out/phase_clocks.log
