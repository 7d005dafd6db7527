#!/bin/bash
# priority of the side stream (the tail DP classes): low (default) / normal / high, two batches in flight
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
for pr in low normal high; do
HLALA_SIDE_PRIORITY=$pr timeout 600 python bench.py --steps 8 --warmup 2 --host-steps 0 --e2e-pairs 0 --no-cpu-baseline --no-extras 2>/dev/null | grep '^{' | tail -1 > gpurun_out/r3_prio_$pr.json
python - $pr <<'PY'
import json, sys
try:
    d = json.load(open('gpurun_out/r3_prio_%s.json' % sys.argv[1]))
    sm = d["config"]["stage_ms"]
    print("[side priority %s] value %d ms %.1f | %s" % (sys.argv[1], d["value"], d["ms_per_step"], {k: round(v, 1) for k, v in sm.items()}))
except Exception as e:
    print("failed", e)
PY
done
