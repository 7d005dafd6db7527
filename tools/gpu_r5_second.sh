#!/bin/bash
# round 5: after the stitch rewrite -- parity tests of the pipeline incl. two batches in flight, the parity sweep, then kernel stats of the bench at 1 M pairs
set -u
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_extend.py tests/test_gpu_align.py tests/test_graph_m.py tests/test_parity_sweep.py tests/test_unpaired.py -x -q -m gpu 2>&1 | tail -8 | tee gpurun_out/r5_second_tests.log
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_r5b
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r5b -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --long-reads 0 --no-extras > $R/gpurun_out/prof_r5b.log 2>&1
find $R/gpurun_out/prof_r5b -name "*kernel_trace.csv" -delete
tail -1 $R/gpurun_out/prof_r5b.log | cut -c1-600
