#!/bin/bash
# round 6: the two-track band kernels -- parity (alignment suite, Graph M, the sweep), then the class statistics of a Graph M batch with and without them
set -u
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1800 python -m pytest tests/test_gpu_extend.py tests/test_gpu_align.py tests/test_graph_m.py -x -q -m gpu 2>&1 | tail -15 | tee gpurun_out/r6_band2_tests.log
timeout 1800 python -m pytest tests/test_parity_sweep.py -x -q -m gpu 2>&1 | tail -8 | tee -a gpurun_out/r6_band2_tests.log
for v in 1 0; do
  echo "== HLALA_DP_BAND2=$v"
  HLALA_DP_BAND2=$v timeout 600 python tools/band_stats.py 1048576 5000000 2>&1 | grep -E "band|16-lane|later|stages"
done
