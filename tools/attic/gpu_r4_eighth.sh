#!/bin/bash
# round 4, eighth session: division-free x/z order (parity + A/B), the host thread's time in the boundary loop, three batches in flight
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
python tools/xz_order_check.py
bash tools/gpu_r4_ab.sh --parity "xz-order:HLALA_X=1"
for nf in 2 3; do
  timeout 900 python bench.py --steps 8 --warmup 2 --no-extras --no-cpu-baseline --in-flight $nf 2>gpurun_out/r4_b_err.log | grep '^{' | tail -1 > gpurun_out/r4_boundary_$nf.json
  python - $nf <<'PY'
import json, sys
d = json.load(open('gpurun_out/r4_boundary_%s.json' % sys.argv[1]))
h = d["host_inclusive"]
print("in flight %s: value %d ms %.1f | resident %d ms %.1f | host thread ms per call %s | pageable %s" % (sys.argv[1], d["value"], d["ms_per_step"], d["config"]["resident"]["value"], d["config"]["resident"]["ms_per_step"],
      {k: round(v, 1) for k, v in h["host_thread_ms_per_call"].items()}, h.get("pageable")))
PY
done
