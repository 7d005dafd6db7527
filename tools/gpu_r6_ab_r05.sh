#!/bin/bash
# round 6: the boundary figure of this build against round 5's final build (a worktree of commit ce88412 under gpurun_in_r05/, built there) on ONE box, alternating
set -u
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd $GRAFT_REPO_ROOT
one() {
  ( cd $1 && timeout 900 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | grep '^{' | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(' $2: boundary %.2f ms/step, resident %.2f' % (d['ms_per_step'], d['config']['resident']['ms_per_step']), {k: round(v, 1) for k, v in d['host_inclusive']['host_thread_ms_per_call'].items()})" )
}
for i in 1 2 3; do one gpurun_in_r05 r05; one . r06; done
