#!/bin/bash
# projection: time on the gene-window and the mixed workload with and without k_rethread_chains (HLALA_RETHREAD=0: the wave-wide form inside k_project_chains)
set -u
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
L=gpurun_out/r3_project_ab.log
echo "== $(date) quick ${1:-}" | tee -a $L
for rg in 1 0; do
for cfg in "262144 5000000 m 1.0" "1048576 5000000 m 0.3"; do
  echo "-- HLALA_RETHREAD=$rg $cfg" | tee -a $L
  ( HLALA_RETHREAD=$rg timeout 900 python tools/dbg_timing.py $cfg 2>&1 | grep -E "^ms " ) | tee -a $L
done
done
