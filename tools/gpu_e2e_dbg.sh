#!/bin/bash
# end-to-end line with the decoder's per-round clocks (HLALA_BAM_DEBUG=1)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
HLALA_BAM_DEBUG=1 timeout 900 python bench.py --steps 2 --warmup 1 --host-steps 0 --no-cpu-baseline  > gpurun_out/r3_bench_e2e_dbg.log 2>&1
python - <<'PY'
import json
l=[x for x in open('gpurun_out/r3_bench_e2e_dbg.log') if x.startswith('{')]
if l:
    d=json.loads(l[-1]); e=d.get("end_to_end") or {}
    print("end_to_end", {k: e.get(k) for k in ("value","decode_s","alignment_and_typing_s","error")})
    for ln in e.get("log", []): print(ln)
PY
