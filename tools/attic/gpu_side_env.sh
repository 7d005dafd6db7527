#!/bin/bash
# bench.py under environment variants (one box, one build): gpu_side_env.sh "<VAR=x bench args>" ...
set -u
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
make -s -C tools/graphm 2>&1 | tail -1
for V in "$@"; do
  ENVS=""; ARGS=""
  for tok in $V; do case "$tok" in *=*) ENVS="$ENVS $tok";; *) ARGS="$ARGS $tok";; esac; done
  ( env $ENVS timeout 900 python bench.py --no-cpu-baseline --no-extras $ARGS 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); c = d['config']
print('[$V]', round(d['value']), 'pairs/s', round(d['ms_per_step'], 1), 'ms', {k: round(v, 1) for k, v in c['stage_ms'].items()}, 'errors', c['chain_errors'])" ) 2>&1 | tee -a gpurun_out/side_env.log
done
