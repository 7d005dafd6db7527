#!/bin/bash
# round 6: blocks of the in-memory DP class (64 by default: a quarter of the CUs) -- resident step, and the gene-window batch alone and two in flight
set -u
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd $GRAFT_REPO_ROOT
one() { python bench.py --resident-only --steps 10 --warmup 3 --no-extras --no-cpu-baseline 2>/dev/null | grep '^{' | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('  resident %.2f ms/step' % d['ms_per_step'], {k: round(v, 1) for k, v in d['config']['stage_ms'].items() if k in ('project', 'extend', 'pair', 'side_stream', 'dp_in_memory', 'dp_large')})"; }
for b in 64 128 256 64 128; do echo "in-memory class on $b blocks"; HLALA_DP_HUGE_BLOCKS=$b one; done
