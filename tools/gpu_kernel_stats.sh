#!/bin/bash
set -u
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_q -- python3 $R/bench.py --pairs 262144 --levels 5000000 --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_q.log 2>&1
find $R/gpurun_out/prof_q -name "*kernel_trace.csv" -delete
cat $R/gpurun_out/prof_q/*/*kernel_stats.csv | cut -c1-200 | head -20
tail -1 $R/gpurun_out/prof_q.log | cut -c1-1500
