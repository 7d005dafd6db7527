#!/bin/bash
# round 5: how the boundary's figure depends on the warm-up and the number of timed steps (one box, back to back)
set -u
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd $GRAFT_REPO_ROOT
for cfg in "6 2" "6 8" "20 5" "40 5" "6 2"; do
  set -- $cfg
  timeout 900 python bench.py --steps $1 --warmup $2 --no-extras --no-cpu-baseline --long-reads 0 2>/dev/null | tail -1 > /tmp/w.json
  python - $1 $2 <<'PY'
import json, sys
d = json.load(open('/tmp/w.json'))
print("steps %s warmup %s: boundary %.2f ms/step, resident %.2f; host ms per call %s" % (sys.argv[1], sys.argv[2], d['ms_per_step'], d['config']['resident']['ms_per_step'], {k: round(v, 1) for k, v in d['host_inclusive']['host_thread_ms_per_call'].items()}))
PY
done
