#!/bin/bash
# round 6: the boundary figure (20 steps, as the driver runs it, without the extras) -- default twice, tail pool 2 and 3, side-stream classes behind the pairing pass
set -u
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
run() {
  echo "== $*"
  env "$@" timeout 900 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline ${BARGS:-} 2>/dev/null | grep '^{' | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(' boundary %.2f ms/step, resident %.2f' % (d['ms_per_step'], d['config']['resident']['ms_per_step']), d['host_inclusive']['host_thread_ms_per_call'])"
}
run A=1
run A=2
BARGS="--tail-pool 2" run A=3
BARGS="--tail-pool 3" run A=4
run HLALA_SIDE_AFTER_PAIR=1
run A=5
