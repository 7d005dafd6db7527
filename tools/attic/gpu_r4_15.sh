#!/bin/bash
# round 4, session 15: boundary loop with the alignment of batch i+2 queued before batch i is read back (three sets of output arrays alive), pool caps
# (bench.py --launch-first and HLALA_POOL_GB existed for this session only: the order lost, profiles/r04_experiments.txt 15)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
run() {
  label="$1"; shift
  env "$@" timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --resident-steps 4 $FLAGS 2>gpurun_out/r4_15_err.log | grep '^{' | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); c = d['config']
print('[$label] boundary %d ms %.1f | resident %d ms %.1f | host' % (d['value'], d['ms_per_step'], c['resident']['value'], c['resident']['ms_per_step']), {k: round(v, 1) for k, v in d['host_inclusive']['host_thread_ms_per_call'].items()})
" || tail -5 gpurun_out/r4_15_err.log
}
FLAGS="" run "read back first" HLALA_X=1
FLAGS="--launch-first" run "launch first, pool 96 GB" HLALA_X=1
FLAGS="--launch-first" run "launch first, pool 170 GB" HLALA_POOL_GB=170
FLAGS="" run "read back first again" HLALA_X=1
rocm-smi --showmeminfo vram 2>/dev/null | tail -4
