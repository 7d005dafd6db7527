#!/bin/bash
# round 4, session 13: the tail classes start right after the first DP class (early sweep) -- parity, A/B, gene-window / backbone batches alone
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
bash tools/gpu_r4_ab.sh --parity "early-tail:HLALA_EARLY_TAIL=1" "late-tail:HLALA_EARLY_TAIL=0"
timeout 900 python -m pytest tests/test_full_scale.py tests/test_unpaired.py tests/test_distributed_gpu.py -m gpu -q -x 2>&1 | tail -3
for e in 1 0; do
HLALA_EARLY_TAIL=$e timeout 900 python bench.py --steps 4 --warmup 2 --resident-steps 4 --long-reads 0 --e2e-pairs 0 --no-cpu-baseline 2>/dev/null | grep '^{' | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); c = d['config']
print('early=$e boundary %d ms %.1f resident %d | gene %d backbone %d' % (d['value'], d['ms_per_step'], c['resident']['value'], c['gene_window_pairs']['pairs_per_s'], c['backbone_pairs']['pairs_per_s']))
"
done
( timeout 1200 python tools/stress_parity.py 4000 2>&1 | tail -3 )
