#!/bin/bash
# quick parity + timing of the lane-per-DP class (single batch, then two in flight)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_extend.py tests/test_gpu_align.py tests/test_graph_m.py -m gpu -q -x > gpurun_out/r3_pytest_h.log 2>&1
echo "pytest H rc=$?" >> gpurun_out/r3_pytest_h.log
tail -6 gpurun_out/r3_pytest_h.log
if ! grep -q "pytest H rc=0" gpurun_out/r3_pytest_h.log; then exit 1; fi
for mode in "--single-batch" ""; do
timeout 600 python bench.py --steps 8 --warmup 2 --host-steps 0 --e2e-pairs 0 --no-cpu-baseline --no-extras $mode 2>/dev/null | grep '^{' | tail -1 > gpurun_out/r3_lane_b$mode.json
python - "$mode" <<'PY'
import json, sys
try:
    d = json.load(open('gpurun_out/r3_lane_b%s.json' % sys.argv[1]))
    sm = d["config"]["stage_ms"]
    print("%s value %d ms %.1f | %s | entering %s" % (sys.argv[1] or "two-in-flight", d["value"], d["ms_per_step"], {k: round(v, 1) for k, v in sm.items()}, d["config"]["dp_calls_entering_class"]))
except Exception as e:
    print("failed", e)
PY
done
