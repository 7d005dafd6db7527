#!/bin/bash
# round 6: the whole GPU suite, then the end-to-end records (one sample; two and four samples taking turns on shared contexts), then the eight-rank dry run
set -u
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -12 | tee gpurun_out/r6_pytest_gpu.txt
SAMPLES="2,4" bash tools/gpu_r6_e2e2.sh 2>&1 | tail -8
bash tools/gpu_r6_multirank.sh 2>&1 | tail -6
