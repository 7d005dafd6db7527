#!/bin/bash
# build variants of the tail classes (threads per DP) on one box and time each: two batches in flight and one batch at a time
#   gpu_tail_variants.sh "<flags1>" "<flags2>" ...
set -u
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
i=0
for EX in "$@"; do
  i=$((i+1)); rm -rf /tmp/v$i && mkdir /tmp/v$i && cp -r hla-la_amd include tools tests oracle bench.py __graft_entry__.py /tmp/v$i/
  ( cd /tmp/v$i && rm -rf hla-la_amd/csrc/_obj/hlala_api.o && make -s -C hla-la_amd/csrc ../libhlala_gpu.so EXTRA="$EX" 2>&1 | grep -E "rror" ) &
done
wait
i=0
for EX in "$@"; do
  i=$((i+1))
  for mode in "" "--single-batch"; do
    ( cd /tmp/v$i && timeout 500 python bench.py --steps 8 --warmup 2 --host-steps 0 --e2e-pairs 0 --no-cpu-baseline --no-extras $mode 2>/dev/null | grep '^{' | tail -1 > /tmp/v$i/out.json
      python - "$EX" "$mode" <<'PY'
import json, sys
try:
    d = json.load(open('out.json'))
    sm = d["config"]["stage_ms"]
    print("[%s] %s value %d ms %.1f | side %.0f 16lane %.0f project %.0f broad %.0f large %.0f huge %.0f wide %.0f" % (sys.argv[1], sys.argv[2] or "two-in-flight", d["value"], d["ms_per_step"], sm["side_stream"], sm["dp_16lane"], sm["project"], sm["dp_broad"], sm["dp_large"], sm["dp_in_memory"], sm["dp_wide"]))
except Exception as e:
    print("[%s] %s failed: %r" % (sys.argv[1], sys.argv[2], e))
PY
    ) | tee -a gpurun_out/r3_tail_variants.log
  done
done
