#!/bin/bash
set -u
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd $GRAFT_REPO_ROOT
make -C oracle 2>&1 | tail -1
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
( time timeout 1500 python bench.py ) > gpurun_out/bench_r01i_full.log 2>&1
tail -5 gpurun_out/bench_r01i_full.log | cut -c1-600
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r01i -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_r01i.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_fetch_d -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $R/gpurun_out/pmc_fetch_d.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_write_d -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $R/gpurun_out/pmc_write_d.log 2>&1
find $R/gpurun_out -name "*kernel_trace.csv" -size +20M -delete
find $R/gpurun_out/prof_r01i $R/gpurun_out/pmc_fetch_d $R/gpurun_out/pmc_write_d -name "*.csv" | head -20
