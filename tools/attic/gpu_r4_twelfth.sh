#!/bin/bash
# round 4, twelfth session: how far beyond its read bases a jump-free call is taken to reach (HLALA_DP_JF_MARGIN)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
for m in 48 16 8 4; do
  bash tools/gpu_r4_ab.sh --modes single "margin$m:HLALA_DP_JF_MARGIN=$m"
  python3 - <<'PY'
import json
d = json.load(open('gpurun_out/r4_ab_last.json'))
print(d["config"]["dp_calls_entering_class"])
PY
done
HLALA_DP_JF_MARGIN=8 timeout 900 python -m pytest tests/test_gpu_extend.py tests/test_gpu_align.py tests/test_graph_m.py -m gpu -q -x 2>&1 | tail -3
