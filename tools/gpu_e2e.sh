#!/bin/bash
# end-to-end only: bench.py with a short resident loop, no host loop, no CPU baseline
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
for fg in "$@"; do
timeout 1200 python bench.py --steps 2 --warmup 1 --host-steps 0 --no-cpu-baseline --e2e-frac-gene $fg > gpurun_out/r3_bench_e2e_$fg.log 2>&1
python - $fg <<'PY'
import json, sys
l=[x for x in open('gpurun_out/r3_bench_e2e_%s.log' % sys.argv[1]) if x.startswith('{')]
if l:
    d=json.loads(l[-1]); e=d.get("end_to_end"); print(sys.argv[1], json.dumps(e)[:2500])
else:
    print(open('gpurun_out/r3_bench_e2e_%s.log' % sys.argv[1]).read()[-2000:])
PY
done
