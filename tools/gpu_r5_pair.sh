#!/bin/bash
# round 5: k_pair_chains rewrite -- parity (alignment tests, sweep, Graph M, long reads), then the phase clocks of the timing build
set -u
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1800 python -m pytest tests/test_gpu_align.py tests/test_parity_sweep.py tests/test_graph_m.py tests/test_long_reads_full.py -m gpu -x -q > gpurun_out/r5_pair_pytest.log 2>&1
tail -4 gpurun_out/r5_pair_pytest.log
timeout 600 python tools/band_stats.py 1048576 5000000 2>&1 | grep -E "stages"
bash tools/gpu_r5_pairtiming.sh
