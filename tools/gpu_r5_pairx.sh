#!/bin/bash
# round 5: what k_pair_chains waits for: builds that stop early (wrong results: timing only)
set -u
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd $GRAFT_REPO_ROOT
run() { timeout 600 python tools/band_stats.py 1048576 5000000 2>&1 | grep -E "stages"; }
for x in ${PAIRX_LIST:-"PAIR_X_LEVEL=1" "PAIR_X_LEVEL=2"}; do
  touch hla-la_amd/csrc/kernel_pair.hip
  make -C hla-la_amd/csrc EXTRA="-D$x" 2>&1 | grep -E " error" | head
  echo "== build -D$x"; run
done
