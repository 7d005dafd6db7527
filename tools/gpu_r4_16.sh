#!/bin/bash
# round 4, session 16: samples of one call taking turns on one device (decode beside alignment): the host program's tests, then two samples of 8.4 M pairs in one call
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
make -s -C oracle 2>&1 | tail -1; make -s -C tools/graphm 2>&1 | tail -1
timeout 1500 python -m pytest tests/test_hla_la_binary.py tests/test_graph_m.py tests/test_end_to_end.py -m gpu -q -x > gpurun_out/r4_16_tests.log 2>&1
echo "tests rc=$?"; tail -4 gpurun_out/r4_16_tests.log
HLALA_HOST_DEBUG=1 timeout 2400 python bench.py --steps 2 --warmup 1 --resident-steps 0 --long-reads 0 --no-cpu-baseline --no-extras-but-e2e --e2e-threads 0 --e2e-samples ${SAMPLES:-2} > gpurun_out/r4_16_e2e.log 2> gpurun_out/r4_16_e2e.err
echo "bench rc=$?"
python3 - <<'PY'
import json
l = [x for x in open('gpurun_out/r4_16_e2e.log') if x.startswith('{')]
if l:
    e = json.loads(l[-1]).get("end_to_end", {})
    print("one sample", {k: e.get(k) for k in ("value", "decode_s", "alignment_and_typing_s", "process_wall_s", "whole_process_pairs_per_s", "error")})
    print("several", e.get("several_samples_one_gpu"))
else:
    print(open('gpurun_out/r4_16_e2e.err').read()[-2000:])
PY
