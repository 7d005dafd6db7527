#!/bin/bash
# rocprofv3 kernel statistics + PMC passes of the default bench workload -> gpurun_out/<tag>_*; `python tools/derive_profiles.py <tag>` turns them into profiles/<tag>_*
#   gpu_profile.sh <tag>        (counters are collected in their own runs, with --kernel-trace only)
set -u
: "${GRAFT_REPO_ROOT:?run through gpurun}"
R=$GRAFT_REPO_ROOT
TAG=${1:-r04}
cd "$R"
mkdir -p gpurun_out
make -s -C tools/graphm 2>&1 | tail -1
rm -rf gpurun_out/${TAG}_stats gpurun_out/${TAG}_fetch gpurun_out/${TAG}_write gpurun_out/${TAG}_sq
cd /tmp && export TMPDIR=/tmp
B="--no-cpu-baseline --resident-only"
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_stats -- python3 $R/bench.py $B > $R/gpurun_out/${TAG}_stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_fetch -- python3 $R/bench.py --steps 1 --warmup 0 --single-batch $B > $R/gpurun_out/${TAG}_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_write -- python3 $R/bench.py --steps 1 --warmup 0 --single-batch $B > $R/gpurun_out/${TAG}_write.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_sq -- python3 $R/bench.py --steps 1 --warmup 0 --single-batch $B > $R/gpurun_out/${TAG}_sq.log 2>&1
find $R/gpurun_out/${TAG}_stats $R/gpurun_out/${TAG}_fetch $R/gpurun_out/${TAG}_write $R/gpurun_out/${TAG}_sq -name "*kernel_trace.csv" -size +8M -delete
tail -1 $R/gpurun_out/${TAG}_stats.log | cut -c1-400
find $R/gpurun_out/${TAG}_stats $R/gpurun_out/${TAG}_fetch $R/gpurun_out/${TAG}_write $R/gpurun_out/${TAG}_sq -name "*.csv" | head -20
