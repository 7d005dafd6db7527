#!/bin/bash
# round 5: linear steps with parallel edges -- the band parity test, the 48-world sweep, and the class statistics of one 262 k / 1 M-pair Graph M batch
set -u
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_align.py tests/test_parity_sweep.py -m gpu -x -q -k "band or sweep or graph_m" > gpurun_out/r5_par_pytest.log 2>&1
tail -5 gpurun_out/r5_par_pytest.log
for p in 262144 1048576; do
  echo "== pairs $p"
  timeout 600 python tools/band_stats.py $p 5000000 2>&1 | grep -E "band:|16-lane|later|stages|fail-over"
done
timeout 900 python bench.py --steps 10 --warmup 4 2>&1 | tail -1 > gpurun_out/r5_par_bench.json; cut -c1-600 gpurun_out/r5_par_bench.json
