#!/bin/bash
# round 2, first GPU call: full GPU test suite incl. the Graph M parity tests, timing per file
set -u
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
make -s -C oracle 2>&1 | tail -1
make -s -C tools/graphm 2>&1 | tail -1
( time timeout 900 python -m pytest tests/test_graph_m.py -x -q -m gpu ) > gpurun_out/r2_graphm.log 2>&1
tail -30 gpurun_out/r2_graphm.log
( time timeout 1500 python -m pytest tests -x -q -m gpu --deselect tests/test_graph_m.py ) > gpurun_out/r2_gpu_tests.log 2>&1
tail -8 gpurun_out/r2_gpu_tests.log
