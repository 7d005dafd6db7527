#!/bin/bash
# round 6: k_stitch_chains' ordered FP64 sum with v_readlane instead of a shuffle per lane step (this build) against the build before (gpurun_in_ab/base.so), on one box:
# the resident step with its stitch stage, the long reads' pad + score stage; then the tests that compare likelihoods
set -u
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd $GRAFT_REPO_ROOT
one() { python bench.py --resident-only --steps 10 --warmup 3 --no-extras --no-cpu-baseline 2>/dev/null | grep '^{' | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('  resident %.2f ms/step' % d['ms_per_step'], {k: round(v, 1) for k, v in d['config']['stage_ms'].items() if k in ('project', 'extend', 'pair', 'side_stream')})"; }
for i in 1 2; do
  echo base; HLALA_LIB_PATH=$GRAFT_REPO_ROOT/gpurun_in_ab/base.so one
  echo new; one
done
echo base; HLALA_LIB_PATH=$GRAFT_REPO_ROOT/gpurun_in_ab/base.so timeout 900 python tools/long_phase.py 50000 5000000 2>&1 | grep "reads/s"
echo new; timeout 900 python tools/long_phase.py 50000 5000000 2>&1 | grep "reads/s"
timeout 1500 python -m pytest tests/test_gpu_align.py tests/test_unpaired.py tests/test_long_reads_full.py -x -q -m gpu 2>&1 | tail -2
