#!/bin/bash
# A/B on one box: the round-1 tree (gpurun_in_r01/, exported from git) against the current tree, round-1 workload
# usage: gpu_ab_r01.sh [pairs] [extra make flags, e.g. -DHLALA_DP_TIMING]
set -u
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd "$GRAFT_REPO_ROOT"
N=${1:-1048576}; EX=${2:-}
rm -rf /tmp/a /tmp/b && cp -r gpurun_in_r01 /tmp/a && mkdir /tmp/b && cp -r hla-la_amd include tools tests __graft_entry__.py /tmp/b/
( cd /tmp/a && touch hla-la_amd/csrc/hlala_api.hip && make -s -C hla-la_amd/csrc ../libhlala_gpu.so EXTRA="$EX" 2>&1 | grep -E "rror" )
( cd /tmp/b && touch hla-la_amd/csrc/hlala_api.hip && make -s -C hla-la_amd/csrc ../libhlala_gpu.so EXTRA="$EX" 2>&1 | grep -E "rror" )
for i in 1; do
  echo "A (round 1):"; ( cd /tmp/a && timeout 600 python tools/dbg_timing.py $N 5000000 2>&1 | tail -4 )
  echo "B (now):"; ( cd /tmp/b && timeout 600 python tools/dbg_timing.py $N 5000000 2>&1 | tail -4 )
done
