#!/bin/bash
# round 6: kernel trace of the resident loop by queue, tail pool 1 and 4 (tools/trace_timeline.py)
set -u
: "${GRAFT_REPO_ROOT:?run through gpurun}"
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for k in ${POOLS:-1 4}; do
  rm -rf $R/gpurun_out/prof_pool
  rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_pool -- python3 $R/bench.py --steps 12 --warmup 6 --resident-only --no-cpu-baseline --long-reads 0 --no-extras --tail-pool $k ${BENCH_ARGS:-} > $R/gpurun_out/r6_pooltrace_$k.log 2>&1
  tail -1 $R/gpurun_out/r6_pooltrace_$k.log | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('tail pool $k: resident ms/step %.2f' % d['ms_per_step'])"
  python3 $R/tools/trace_timeline.py $R/gpurun_out/prof_pool ${WIN:-800} 1.0 > $R/gpurun_out/r6_pooltrace_$k.txt
  head -120 $R/gpurun_out/r6_pooltrace_$k.txt
  rm -rf $R/gpurun_out/prof_pool
done
