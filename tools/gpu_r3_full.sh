#!/bin/bash
# the driver's round-end sequence: the whole -m gpu suite, smoke(), the default bench
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
( time timeout 2400 python -m pytest tests -m gpu -q -x --durations=12 ) > gpurun_out/r3_pytest_full.log 2>&1
echo "pytest full rc=$?" >> gpurun_out/r3_pytest_full.log
tail -22 gpurun_out/r3_pytest_full.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
( time timeout 1500 python bench.py ) > gpurun_out/r3_bench_full.log 2> gpurun_out/r3_bench_full.err
tail -4 gpurun_out/r3_bench_full.err
python - <<'PY'
import json
l=[x for x in open('gpurun_out/r3_bench_full.log') if x.startswith('{')]
if l:
    d=json.loads(l[-1])
    print("value", d["value"], "ms", d["ms_per_step"], "roofline", {k: d["roofline"][k] for k in ("kernel","kernel_ms","achieved","frac","traffic")})
    print("host_inclusive", json.dumps(d.get("host_inclusive"))[:600])
    e=d.get("end_to_end") or {}
    print("end_to_end", {k: e.get(k) for k in ("value","pairs","decode_s","decode_threads","alignment_and_typing_s","typing_phases","error")})
    print("cpu", d.get("cpu_baseline",{}).get("value"), "extras_error", d["config"].get("extras_error"))
PY
