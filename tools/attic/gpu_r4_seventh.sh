#!/bin/bash
# round 4, seventh session: profile artefacts (kernel stats + PMC passes), the N > 1 dry runs, the decoder at several thread counts
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
bash tools/gpu_profile.sh r04 2>&1 | tail -12 | cut -c1-300
bash tools/gpu_multirank_dryrun.sh 2>&1 | tail -14 | cut -c1-900
timeout 1200 python bench.py --steps 2 --warmup 1 --resident-steps 0 --long-reads 0 --no-cpu-baseline --e2e-threads 128,16,24,32,48,64,96 > gpurun_out/r4_e2e_threads.log 2> gpurun_out/r4_e2e_threads.err
python - <<'PY'
import json
l = [x for x in open('gpurun_out/r4_e2e_threads.log') if x.startswith('{')]
if l:
    e = json.loads(l[-1]).get("end_to_end") or {}
    rows = [e] + list(e.get("other_thread_counts") or [])
    for r in rows:
        print({k: r.get(k) for k in ("decode_threads", "decode_s", "value", "alignment_and_typing_s", "process_wall_s", "error") if k in r})
else:
    print(open('gpurun_out/r4_e2e_threads.err').read()[-1500:])
PY
