#!/bin/bash
# round 6: the host program -- several samples taking turns on one device (decode of sample k + 1 beside the alignment of sample k), the communicator, the tail pool in the walk;
# then the end-to-end records of the bench (one sample; two and four samples in one call)
set -u
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_hla_la_binary.py tests/test_comm.py tests/test_end_to_end.py -x -q -m gpu 2>&1 | tail -12 | tee gpurun_out/r6_e2e_tests.log
timeout 2400 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --long-reads 0 --no-extras-but-e2e --resident-steps 0 --e2e-threads 0 2>gpurun_out/r6_e2e_bench.err | grep '^{' | tail -1 > gpurun_out/r6_e2e_bench.json
python - <<'PY'
import json
j = json.load(open("gpurun_out/r6_e2e_bench.json"))
e = j.get("end_to_end", {})
print("one sample: %s pairs/s (decode %s s, alignment and typing %s s)" % (e.get("value"), e.get("decode_s"), e.get("alignment_and_typing_s")))
for s in e.get("several_samples", []):
    print(json.dumps({k: v for k, v in s.items() if k != "what"})[:1500])
if "error" in e: print(e["error"])
PY
