#!/bin/bash
# round 4, fourth session: the jump-free instantiation of the 16-lane class (parity, A/B), the list of counters rocprofv3 offers on this box
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
bash tools/gpu_r4_ab.sh --parity "jf:HLALA_DP_JF=1" "nojf:HLALA_DP_JF=0"
python3 - <<'PY'
import json
d = json.load(open('gpurun_out/r4_ab_last.json'))
print(d["config"]["dp_calls_entering_class"])
PY
rocprofv3 -L 2>/dev/null | grep -o "TCC_[A-Z0-9_]*\|TCP_[A-Z0-9_]*" | sort -u | tr '\n' ' ' > gpurun_out/r4_counters_tcc.txt
wc -c gpurun_out/r4_counters_tcc.txt
