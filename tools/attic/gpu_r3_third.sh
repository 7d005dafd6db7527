#!/bin/bash
# round 3, third GPU call: several waves per DP in the tail classes -- parity first (short timeouts: a hang must not cost the box), then timing
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 400 python -m pytest tests/test_graph_m.py tests/test_gpu_align.py tests/test_gpu_extend.py -m gpu -q -x > gpurun_out/r3_pytest_d.log 2>&1
echo "pytest D rc=$?" >> gpurun_out/r3_pytest_d.log
tail -15 gpurun_out/r3_pytest_d.log
if ! grep -q "pytest D rc=0" gpurun_out/r3_pytest_d.log; then exit 1; fi
timeout 400 python -m pytest tests/test_full_scale.py -m gpu -q -x > gpurun_out/r3_pytest_e.log 2>&1
echo "pytest E rc=$?" >> gpurun_out/r3_pytest_e.log
tail -8 gpurun_out/r3_pytest_e.log
for mode in "" "--single-batch"; do
timeout 600 python bench.py --steps 10 --warmup 3 --host-steps 0 --e2e-pairs 0 --no-cpu-baseline $mode > gpurun_out/r3_bench_c$mode.log 2>&1
python - "$mode" <<'PY'
import json, sys
l=[x for x in open('gpurun_out/r3_bench_c%s.log' % sys.argv[1]) if x.startswith('{')]
if l:
    d=json.loads(l[-1]); print(sys.argv[1] or "two in flight", "value", round(d["value"]), "ms", round(d["ms_per_step"],1)); print(d["config"]["stage_ms"]); print({k: round(v["pairs_per_s"]) for k, v in d["config"].items() if isinstance(v, dict) and "pairs_per_s" in v})
PY
done
timeout 900 python bench.py --steps 2 --warmup 1 --host-steps 0 --no-extras --no-cpu-baseline > gpurun_out/r3_bench_e2e.log 2>&1
python - <<'PY'
import json
l=[x for x in open('gpurun_out/r3_bench_e2e.log') if x.startswith('{')]
if l:
    d=json.loads(l[-1]); print("end_to_end", json.dumps(d.get("end_to_end"))[:1800])
PY
