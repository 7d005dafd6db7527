#!/bin/bash
set -u
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd $GRAFT_REPO_ROOT
make -C oracle 2>&1 | tail -1
timeout 600 python -m pytest tests/test_gpu_extend.py tests/test_gpu_align.py -x -q -m gpu 2>&1 | tail -3
timeout 300 python tools/dbg_timing.py 262144 5000000 2>&1 | tail -2
timeout 300 python tools/dbg_timing.py 262144 5000000 2>&1 | tail -2
