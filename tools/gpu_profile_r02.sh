#!/bin/bash
# rocprofv3 kernel statistics + PMC passes of the default bench workload -> gpurun_out/r02_*; tools/derive_profiles.py turns them into profiles/r02_*
set -u
: "${GRAFT_REPO_ROOT:?run through gpurun}"
R=$GRAFT_REPO_ROOT
cd "$R"
mkdir -p gpurun_out
make -s -C tools/graphm 2>&1 | tail -1
rm -rf gpurun_out/r02_stats gpurun_out/r02_fetch gpurun_out/r02_write gpurun_out/r02_sq
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r02_stats -- python3 $R/bench.py --no-cpu-baseline --no-extras > $R/gpurun_out/r02_stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/r02_fetch -- python3 $R/bench.py --steps 1 --warmup 0 --single-batch --no-cpu-baseline --no-extras > $R/gpurun_out/r02_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/r02_write -- python3 $R/bench.py --steps 1 --warmup 0 --single-batch --no-cpu-baseline --no-extras > $R/gpurun_out/r02_write.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $R/gpurun_out/r02_sq -- python3 $R/bench.py --steps 1 --warmup 0 --single-batch --no-cpu-baseline --no-extras > $R/gpurun_out/r02_sq.log 2>&1
find $R/gpurun_out/r02_stats $R/gpurun_out/r02_fetch $R/gpurun_out/r02_write $R/gpurun_out/r02_sq -name "*kernel_trace.csv" -size +8M -delete
tail -1 $R/gpurun_out/r02_stats.log | cut -c1-400
find $R/gpurun_out/r02_stats $R/gpurun_out/r02_fetch $R/gpurun_out/r02_write $R/gpurun_out/r02_sq -name "*.csv" | head -20
