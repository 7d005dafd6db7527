#!/bin/bash
# round 3, second GPU call: the tests changed since the first call, then the full default bench (host_inclusive loop, end_to_end through HLA-LA)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_hla_la_binary.py tests/test_distributed_gpu.py tests/test_end_to_end.py -m gpu -q -x > gpurun_out/r3_pytest_c.log 2>&1
echo "pytest C rc=$?" >> gpurun_out/r3_pytest_c.log
tail -25 gpurun_out/r3_pytest_c.log
timeout 1500 python bench.py > gpurun_out/r3_bench_b.log 2> gpurun_out/r3_bench_b.err
echo "bench rc=$?" >> gpurun_out/r3_bench_b.err
tail -5 gpurun_out/r3_bench_b.err
python - <<'PY'
import json
l=[x for x in open('gpurun_out/r3_bench_b.log') if x.startswith('{')]
if l:
    d=json.loads(l[-1])
    print("value", d["value"], "ms", d["ms_per_step"])
    print("host_inclusive", json.dumps(d.get("host_inclusive"))[:1500])
    print("end_to_end", json.dumps(d.get("end_to_end"))[:2500])
    print("extras_error", d["config"].get("extras_error"))
PY
