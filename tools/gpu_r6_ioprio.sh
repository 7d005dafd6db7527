#!/bin/bash
# round 6: priority of the upload / reader streams (highest by default since round 6; HLALA_IO_PRIORITY=normal: as before) -- the boundary loop, whose one host thread waits for the
# small kernels of hlala_batch_create and of the read-back beside the persistent kernels
set -u
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd $GRAFT_REPO_ROOT
one() { python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline --resident-steps 4 2>/dev/null | grep '^{' | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('  boundary %.2f ms/step, resident %.2f' % (d['ms_per_step'], d['config']['resident']['ms_per_step']), {k: round(v, 1) for k, v in d['host_inclusive']['host_thread_ms_per_call'].items()})"; }
for i in 1 2 3; do
  echo "io streams: normal priority"; HLALA_IO_PRIORITY=normal one
  echo "io streams: highest priority"; one
done
