#!/bin/bash
set -u
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $R/gpurun_out/pmc_sq -- python3 $R/bench.py --pairs 262144 --levels 5000000 --steps 1 --warmup 0 --no-cpu-baseline > $R/gpurun_out/pmc_sq.log 2>&1
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_FLAT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d $R/gpurun_out/pmc_sq2 -- python3 $R/bench.py --pairs 262144 --levels 5000000 --steps 1 --warmup 0 --no-cpu-baseline > $R/gpurun_out/pmc_sq2.log 2>&1
find $R/gpurun_out/pmc_sq $R/gpurun_out/pmc_sq2 -name "*kernel_trace.csv" -delete
tail -2 $R/gpurun_out/pmc_sq.log | cut -c1-300
