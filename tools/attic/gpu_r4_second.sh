#!/bin/bash
# round 4, second session: parity of the position order / sparse item slots / new stitch kernel, A/B against input order, kernel stats, DP timing build
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
bash tools/gpu_r4_ab.sh --parity "locality:HLALA_LOCALITY=1" "input-order:HLALA_LOCALITY=0"
R=$GRAFT_REPO_ROOT
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r4_prof -- python3 $R/bench.py --steps 4 --warmup 1 --resident-only --no-cpu-baseline > $R/gpurun_out/r4_prof.log 2>&1 )
find gpurun_out/r4_prof -name "*kernel_trace.csv" -delete
cat gpurun_out/r4_prof/*/*kernel_stats.csv | cut -c1-160 | head -24
# timing build in a scratch copy
rm -rf /tmp/tb && mkdir -p /tmp/tb && cp -r hla-la_amd include tools tests /tmp/tb/ && cd /tmp/tb
touch hla-la_amd/csrc/hlala_api.hip; make -C hla-la_amd/csrc EXTRA=-DHLALA_DP_TIMING ../libhlala_gpu.so 2>&1 | grep -E "error" 
timeout 600 python tools/dbg_timing.py 1048576 5000000 m 0.3 2>&1 | tail -7 | tee -a $R/gpurun_out/r4_dp_timing.log
