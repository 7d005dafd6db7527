#!/bin/bash
# round 6: k_project_chains<384 columns> compiled for four wavefronts per SIMD (128 registers, 4 spilled; gpurun_in_ab/pwps4.so) with 13 blocks per CU (what its 11.7 KB of LDS allow)
# against the default (153 registers, 12 blocks): the resident step and the projection stage
set -u
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd $GRAFT_REPO_ROOT
one() { python bench.py --resident-only --steps 10 --warmup 3 --no-extras --no-cpu-baseline 2>/dev/null | grep '^{' | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('  resident %.2f ms/step' % d['ms_per_step'], {k: round(v, 1) for k, v in d['config']['stage_ms'].items() if k in ('project', 'extend', 'pair')})"; }
for i in 1 2; do
  echo default; one
  echo "four per SIMD, 13 blocks"; HLALA_LIB_PATH=$GRAFT_REPO_ROOT/gpurun_in_ab/pwps4.so HLALA_PROJ_WAVES=13 one
done
echo "four per SIMD, 12 blocks"; HLALA_LIB_PATH=$GRAFT_REPO_ROOT/gpurun_in_ab/pwps4.so one
