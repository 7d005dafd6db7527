#!/bin/bash
# round 5: one-off stress runs on the final build: eight larger stand-in worlds (15 000 pairs each), a second parity sweep with larger batches (40 worlds x 3 000 pairs, seeds 30000-30039)
set -u
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 1500 python tools/stress_parity.py 15000 ) > gpurun_out/r5_stress_parity.txt 2>&1
tail -10 gpurun_out/r5_stress_parity.txt | cut -c1-250
( timeout 2400 python tools/parity_sweep.py 40 3000 30000 ) > gpurun_out/r5_parity_sweep_large.txt 2>&1
tail -3 gpurun_out/r5_parity_sweep_large.txt | cut -c1-250
