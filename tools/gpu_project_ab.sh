#!/bin/bash
# projection kernel: parity of stage A (chains against the oracle) on the current build, its time on the mixed / gene-window / backbone workloads, phase clocks
set -u
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
L=gpurun_out/r3_project_ab.log
echo "== $(date) ${1:-}" | tee -a $L
( timeout 1500 python -m pytest tests/test_gpu_align.py tests/test_graph_m.py tests/test_unpaired.py -m gpu -x -q 2>&1 | tail -3 ) | tee -a $L
( timeout 600 python tools/stress_parity.py 2000 2>&1 | tail -2 ) | tee -a $L
for cfg in "1048576 5000000 m 0.3" "262144 5000000 m 1.0" "262144 5000000 m 0.0"; do
  echo "-- $cfg" | tee -a $L
  ( HLALA_DEBUG=1 timeout 900 python tools/dbg_timing.py $cfg 2>&1 | grep -E "^ms |project" ) | tee -a $L
done
( timeout 900 python bench.py --steps 6 --warmup 2 --host-steps 0 --e2e-pairs 0 --no-cpu-baseline --no-extras 2>/dev/null | grep '^{' | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); sm=d['config']['stage_ms']
print('bench two-in-flight value %d ms %.1f | %s' % (d['value'], d['ms_per_step'], ' '.join('%s %.1f' % (k, v) for k, v in sm.items())))" ) | tee -a $L
