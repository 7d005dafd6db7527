#!/bin/bash
# round 5: the boundary's copies are shader kernels of the runtime (__amd_rocclr_copyBuffer) -- runtime switches that change how they run
set -u
: "${GRAFT_REPO_ROOT:?run through gpurun}"
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for v in "X=0" "GPU_BLIT_ENGINE_TYPE=2" "DEBUG_CLR_LIMIT_BLIT_WG=8" "DEBUG_CLR_LIMIT_BLIT_WG=32" "GPU_FORCE_BLIT_COPY_SIZE=0"; do
  rm -rf $R/gpurun_out/prof_sdma
  env $v rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_sdma -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras --long-reads 0 > $R/gpurun_out/prof_sdma.log 2>&1
  echo "== $v: $(grep -h '^{' $R/gpurun_out/prof_sdma.log | tail -1 | python3 -c 'import sys,json; j=json.loads(sys.stdin.read()); print(round(j["ms_per_step"],1), "resident", round(j["config"]["resident"]["ms_per_step"],1), j["host_inclusive"]["host_thread_ms_per_call"])' 2>&1 | tail -1)"
  grep -h "rocclr_copyBuffer" $R/gpurun_out/prof_sdma/*/*kernel_stats.csv | cut -d, -f1-4
  find $R/gpurun_out/prof_sdma -name "*kernel_trace.csv" -delete
done
