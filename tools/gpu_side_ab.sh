#!/bin/bash
# bench.py on build variants of the current tree (one box): gpu_side_ab.sh "<EXTRA flags 1>" "<EXTRA flags 2>" ...   (each: two batches in flight, then --single-batch)
set -u
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
make -s -C tools/graphm 2>&1 | tail -1
i=0
for EX in "$@"; do
  i=$((i+1)); rm -rf /tmp/v$i && mkdir /tmp/v$i && cp -r hla-la_amd include tools tests profiles bench.py __graft_entry__.py /tmp/v$i/
  ( cd /tmp/v$i && touch hla-la_amd/csrc/hlala_api.hip && make -s -C hla-la_amd/csrc ../libhlala_gpu.so EXTRA="$EX" 2>&1 | grep -E "rror" )
done
i=0
for EX in "$@"; do
  i=$((i+1))
  for MODE in "--steps 6 --warmup 2" "--single-batch --steps 3 --warmup 1"; do
    ( cd /tmp/v$i && timeout 900 python bench.py --no-cpu-baseline --no-extras $MODE 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); c = d['config']
print('[$EX] [$MODE]', round(d['value']), 'pairs/s', round(d['ms_per_step'], 1), 'ms', {k: round(v, 1) for k, v in c['stage_ms'].items()}, 'errors', c['chain_errors'])" ) 2>&1 | tee -a gpurun_out/side_ab.log
  done
done
