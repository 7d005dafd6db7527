#!/bin/bash
set -u
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
make -s -C tools/graphm 2>&1 | tail -1
make -s -C oracle 2>&1 | tail -1
( timeout 900 python -m pytest tests/test_graph_m.py -x -q -m gpu 2>&1 | tail -5 ) > gpurun_out/r2_graphm.log 2>&1
tail -5 gpurun_out/r2_graphm.log
HLALA_DEBUG=1 timeout 900 python tools/dp_profile.py "$@" > gpurun_out/dp_profile.log 2>&1
head -24 gpurun_out/dp_profile.log
