#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python bench.py --steps 2 --warmup 1 --host-steps 0 --no-cpu-baseline > gpurun_out/r3_bench_e2e_c.log 2>&1
python - <<'PY'
import json
l=[x for x in open('gpurun_out/r3_bench_e2e_c.log') if x.startswith('{')]
if l:
    d=json.loads(l[-1]); e=d.get("end_to_end") or {}
    print("end_to_end", {k: e.get(k) for k in ("value","pairs","decode_s","decode_threads","alignment_and_typing_s","typing_phases","process_wall_s","error")})
PY
timeout 900 python -m pytest tests/test_config3_stream.py tests/test_bam.py tests/test_end_to_end.py -m gpu -q -x -s 2>&1 | grep -E "config 3|passed|failed|Error" | cut -c1-900
