#!/bin/bash
# round 5: kernel stats of the bench at 1 M pairs, two batches in flight (no extras)
set -u
: "${GRAFT_REPO_ROOT:?run through gpurun}"
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_r5b
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r5b -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --long-reads 0 --no-extras > $R/gpurun_out/prof_r5b.log 2>&1
find $R/gpurun_out/prof_r5b -name "*kernel_trace.csv" -delete
tail -1 $R/gpurun_out/prof_r5b.log | cut -c1-400
