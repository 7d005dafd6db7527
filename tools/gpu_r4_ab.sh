#!/bin/bash
# tools/gpu_r4_ab.sh [--parity] [--modes "two single"] "label:ENV=1 ENV2=x" ...   -- resident bench per environment variant on one box (round 4)
# Lines go to gpurun_out/r4_ab.log (appended) and to stdout.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
MODES="two single"
if [ "$1" = "--parity" ]; then
  shift
  make -s -C oracle 2>&1 | tail -1; make -s -C tools/graphm 2>&1 | tail -1
  timeout 900 python -m pytest tests/test_gpu_extend.py tests/test_gpu_align.py tests/test_graph_m.py tests/test_unpaired.py -m gpu -q -x > gpurun_out/r4_parity.log 2>&1
  echo "parity rc=$?" | tee -a gpurun_out/r4_parity.log
  tail -5 gpurun_out/r4_parity.log
fi
if [ "$1" = "--modes" ]; then MODES="$2"; shift; shift; fi
echo "== $(date) $(git rev-parse --short HEAD 2>/dev/null)" >> gpurun_out/r4_ab.log
for v in "$@"; do
  label="${v%%:*}"; envs="${v#*:}"; [ "$envs" = "$v" ] && envs=""
  for mode in $MODES; do
    flag=""; [ "$mode" = single ] && flag="--single-batch"
    env $envs timeout 600 python bench.py --steps 8 --warmup 2 --resident-only --no-cpu-baseline $flag 2>gpurun_out/r4_ab_err.log | grep '^{' | tail -1 > gpurun_out/r4_ab_last.json
    python - "$label" "$mode" <<'PY' | tee -a gpurun_out/r4_ab.log
import json, sys
try:
    d = json.load(open('gpurun_out/r4_ab_last.json'))
    r = d["config"].get("resident", d)
    sm = d["config"]["stage_ms"]
    print("[%s] %s value %d ms %.1f | %s" % (sys.argv[1], sys.argv[2], r["value"], r["ms_per_step"], {k: round(v, 1) for k, v in sm.items() if v}))
except Exception as e:
    print("[%s] %s failed %r" % (sys.argv[1], sys.argv[2], e)); print(open('gpurun_out/r4_ab_err.log').read()[-1500:])
PY
  done
done
