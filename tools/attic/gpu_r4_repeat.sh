#!/bin/bash
# round 4: run-to-run spread of the bench headline on one box (the boundary loop and the resident loop of `bench.py --steps 20 --warmup 5`, extras off)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
for i in 1 2 3 4 5 6; do
  timeout 600 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | grep '^{' | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); c = d['config']
print('run $i: boundary %d pairs/s (%.1f ms per step) | resident %d (%.1f ms) | 16-lane class %.1f ms' % (d['value'], d['ms_per_step'], c['resident']['value'], c['resident']['ms_per_step'], d['roofline']['kernel_ms']))
"
done | tee gpurun_out/r4_bench_repeat.txt
