#!/bin/bash
# timing of build variants of the current tree on one box (round-1 workload): gpu_variants.sh <pairs> "<flags1>" "<flags2>" ...
set -u
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd "$GRAFT_REPO_ROOT"
N=$1; shift
i=0
for EX in "$@"; do
  i=$((i+1)); rm -rf /tmp/v$i && mkdir /tmp/v$i && cp -r hla-la_amd include tools tests __graft_entry__.py /tmp/v$i/
  ( cd /tmp/v$i && touch hla-la_amd/csrc/hlala_api.hip && make -s -C hla-la_amd/csrc ../libhlala_gpu.so EXTRA="$EX" 2>&1 | grep -E "rror" )
done
for rep in 1 2; do i=0; for EX in "$@"; do i=$((i+1)); echo "[$EX]"; ( cd /tmp/v$i && timeout 600 python tools/dbg_timing.py $N 5000000 2>&1 | tail -2 | head -1 ); done; done
