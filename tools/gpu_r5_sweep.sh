#!/bin/bash
# round 5: the parity sweep at its full size on the final build: 120 random worlds (seeds 20000-20119, 600 pairs each; none shared with the 48 under pytest), product against oracle
set -u
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 2400 python tools/parity_sweep.py ${1:-120} ${2:-600} ${3:-20000} > gpurun_out/r5_parity_sweep.txt 2>&1
tail -4 gpurun_out/r5_parity_sweep.txt
