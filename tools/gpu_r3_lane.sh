#!/bin/bash
# lane-per-DP class: parity (short timeouts), then A/B against HLALA_DP_LANE=0
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_extend.py tests/test_gpu_align.py -m gpu -q -x > gpurun_out/r3_pytest_f.log 2>&1
echo "pytest F rc=$?" >> gpurun_out/r3_pytest_f.log
tail -30 gpurun_out/r3_pytest_f.log
if ! grep -q "pytest F rc=0" gpurun_out/r3_pytest_f.log; then exit 1; fi
timeout 600 python -m pytest tests/test_graph_m.py tests/test_hla_la_binary.py tests/test_end_to_end.py tests/test_full_scale.py -m gpu -q -x > gpurun_out/r3_pytest_g.log 2>&1
echo "pytest G rc=$?" >> gpurun_out/r3_pytest_g.log
tail -12 gpurun_out/r3_pytest_g.log
for lane in 1 0; do
for mode in "" "--single-batch"; do
HLALA_DP_LANE=$lane timeout 600 python bench.py --steps 8 --warmup 2 --host-steps 0 --e2e-pairs 0 --no-cpu-baseline --no-extras $mode 2>/dev/null | grep '^{' | tail -1 > gpurun_out/r3_lane_$lane$mode.json
python - $lane "$mode" <<'PY'
import json, sys
try:
    d = json.load(open('gpurun_out/r3_lane_%s%s.json' % (sys.argv[1], sys.argv[2])))
    sm = d["config"]["stage_ms"]
    print("[lane=%s] %s value %d ms %.1f | %s | entering %s" % (sys.argv[1], sys.argv[2] or "two-in-flight", d["value"], d["ms_per_step"], {k: round(v, 1) for k, v in sm.items()}, d["config"]["dp_calls_entering_class"]))
except Exception as e:
    print("failed", e)
PY
done
done
