#!/bin/bash
# round 5: consecutive batches' tail classes on two side streams in turn (HLALA_SIDE_STREAMS=2): parity with two batches in flight, then the resident step and the boundary, A/B
set -u
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
HLALA_SIDE_STREAMS=2 timeout 1500 python -m pytest tests/test_graph_m.py tests/test_config3_stream.py tests/test_full_scale.py -m gpu -x -q > gpurun_out/r5_twosides_pytest.log 2>&1
tail -3 gpurun_out/r5_twosides_pytest.log
for v in 1 2 1 2; do
  echo "== HLALA_SIDE_STREAMS=$v"
  HLALA_SIDE_STREAMS=$v timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --long-reads 0 --no-extras 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(' boundary %.2f ms/step, resident %.2f' % (d['ms_per_step'], d['config']['resident']['ms_per_step']), {k: round(v,1) for k,v in d['config']['stage_ms'].items()})"
done
