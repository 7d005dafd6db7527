#!/bin/bash
# Graph M parity tests on the GPU
set -u
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
make -s -C oracle 2>&1 | tail -1
make -s -C tools/graphm 2>&1 | tail -1
( time timeout 900 python -m pytest tests/test_graph_m.py -x -q -m gpu ) > gpurun_out/r2_graphm.log 2>&1
tail -15 gpurun_out/r2_graphm.log
