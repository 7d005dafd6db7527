#!/bin/bash
# round 5: parity tests of the extension stage, then the class statistics of a Graph M batch
set -u
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_extend.py tests/test_gpu_align.py tests/test_graph_m.py -x -q -m gpu 2>&1 | tail -15 | tee gpurun_out/r5_first_tests.log
MIX_PAIRS=1048576 bash tools/gpu_r5_mix.sh
