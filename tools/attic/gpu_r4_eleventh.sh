#!/bin/bash
# round 4, eleventh session: the 8-lane jump-free instantiation -- parity suite, random stress, A/B against the 16-lane start
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
bash tools/gpu_r4_ab.sh --parity "jf8:HLALA_DP_JF8=1" "nojf8:HLALA_DP_JF8=0"
python3 - <<'PY'
import json
d = json.load(open('gpurun_out/r4_ab_last.json'))
print(d["config"]["dp_calls_entering_class"])
PY
timeout 900 python -m pytest tests/test_full_scale.py tests/test_unpaired.py -m gpu -q -x 2>&1 | tail -3
( timeout 1500 python tools/stress_parity.py 5000 2>&1 | tail -12 ) | tee gpurun_out/r4_stress.log
make -s -C oracle 2>&1 | tail -1
timeout 900 python -m pytest tests/test_hla_la_binary.py tests/test_end_to_end.py -m gpu -q -x 2>&1 | tail -3
timeout 1500 python bench.py --steps 6 --warmup 2 --resident-steps 0 --long-reads 0 --no-cpu-baseline --no-extras-but-e2e --e2e-threads 0 > gpurun_out/r4_e2e_walk.log 2> gpurun_out/r4_e2e_walk.err
python - <<'PY'
import json
l = [x for x in open('gpurun_out/r4_e2e_walk.log') if x.startswith('{')]
if l:
    d = json.loads(l[-1]); e = d.get("end_to_end") or {}
    print("boundary value %d ms %.1f" % (d["value"], d["ms_per_step"]))
    print("e2e", {k: e.get(k) for k in ("value", "decode_s", "page_locking_and_insert_size_s", "alignment_and_typing_s", "window_fill_beside_the_gpu_s", "typing_phases", "error")})
else:
    print(open('gpurun_out/r4_e2e_walk.err').read()[-2000:])
PY
