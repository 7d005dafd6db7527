#!/bin/bash
# round 6: the eight-rank dry run, then the phase clocks of the long-read projection
set -u
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd $GRAFT_REPO_ROOT
bash tools/gpu_r6_multirank.sh 2>&1 | tail -6
HLALA_DEBUG=1 timeout 900 python tools/long_phase.py 8000 5000000 2>&1 | tail -4
