#!/bin/bash
# phase clocks (timing build in a scratch copy) for the mixed workload, gene-window pairs only and backbone pairs only
set -u
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
make -s -C tools/graphm 2>&1 | tail -1
rm -rf /tmp/vt && mkdir /tmp/vt && cp -r hla-la_amd include tools tests __graft_entry__.py /tmp/vt/
( cd /tmp/vt && rm -f hla-la_amd/csrc/_obj/hlala_api.o && make -s -C hla-la_amd/csrc ../libhlala_gpu.so EXTRA="-DHLALA_DP_TIMING" 2>&1 | grep -E "rror" )
for cfg in "1048576 5000000 m 0.3" "262144 5000000 m 1.0" "262144 5000000 m 0.0"; do
  echo "== $cfg" | tee -a gpurun_out/r3_phase_clocks.log
  ( cd /tmp/vt && HLALA_DEBUG=1 timeout 900 python tools/dbg_timing.py $cfg 2>&1 | tail -8 ) | tee -a gpurun_out/r3_phase_clocks.log
done
