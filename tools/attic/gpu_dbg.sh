#!/bin/bash
# HLALA_DEBUG phase clocks: tools/gpu_dbg.sh <args of tools/dbg_timing.py>
set -u
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
make -s -C tools/graphm 2>&1 | tail -1
HLALA_DEBUG=1 timeout 600 python tools/dbg_timing.py "$@" 2>&1 | tail -7
