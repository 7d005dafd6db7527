#!/bin/bash
# round 6: four samples taking turns on one device, decoder threads 0 (default: 32) / 16 / 12 / 8 / 6
set -u
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 3000 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --long-reads 0 --no-extras-but-e2e --resident-steps 0 --e2e-threads 0 --e2e-samples "${SAMPLES:-4:16,4:12,4:8,4:6,2:8}" 2>gpurun_out/r6_e2e2_bench.err | grep '^{' | tail -1 > gpurun_out/r6_e2e2_bench.json
python - <<'PY'
import json
j = json.load(open("gpurun_out/r6_e2e2_bench.json"))
e = j.get("end_to_end", {})
print("one sample: %s pairs/s (decode %s s, alignment and typing %s s)" % (e.get("value"), e.get("decode_s"), e.get("alignment_and_typing_s")))
for s in e.get("several_samples", []):
    print(s.get("samples"), "samples, decode threads", s.get("decode_threads_asked"), ": %.0f pairs/s, wall %.2f s" % (s.get("value", 0), s.get("wall_s", 0)), [l[12:24] + l[l.find("(BAM decode"):l.find(", of which")] for l in s.get("per_sample_lines", [])])
if "error" in e: print(e["error"])
PY
