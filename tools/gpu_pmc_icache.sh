#!/bin/bash
set -u
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -i -o "SQC_[A-Z_]*\|SQ_IFETCH[A-Z_]*\|SQ_INST_LEVEL[A-Z_]*\|SQ_WAIT_INST[A-Z_]*\|SQ_INSTS_BRANCH\|SQ_INSTS_[A-Z_]*" | sort -u | tr '\n' ' ' > $R/gpurun_out/pmc_list.txt
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_WAVE_CYCLES SQ_INSTS_BRANCH --kernel-trace --output-format csv -d $R/gpurun_out/pmc_ic -- python3 $R/bench.py --pairs 262144 --levels 5000000 --steps 1 --warmup 0 --no-cpu-baseline > $R/gpurun_out/pmc_ic.log 2>&1
find $R/gpurun_out/pmc_ic -name "*kernel_trace.csv" -delete
tail -2 $R/gpurun_out/pmc_ic.log | cut -c1-300
ls -R $R/gpurun_out/pmc_ic | head
