#!/bin/bash
# round 5: k_stitch_chains drawing column ROWS (position order; only chains with rows) instead of chain numbers: launch durations of k_stitch_chains / k_pair_chains
# from the kernel trace of the bench's resident loop, and the resident step
set -u
: "${GRAFT_REPO_ROOT:?run through gpurun}"
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for cfg in "0 12" "1 8" "1 12" "1 16" "1 32"; do
  set -- $cfg
  rm -rf $R/gpurun_out/prof_st
  HLALA_STITCH_BY_ROW=$1 HLALA_STITCH_DRAW=$2 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_st -- python3 $R/bench.py --steps 4 --warmup 1 --resident-only --no-cpu-baseline --long-reads 0 --no-extras > $R/gpurun_out/prof_st.log 2>&1
  python3 - $R/gpurun_out/prof_st "$cfg" $R/gpurun_out/prof_st.log <<'PY'
import csv, glob, sys, json
f = glob.glob(sys.argv[1] + '/*/*kernel_trace.csv')[0]
d = {}
for r in csv.DictReader(open(f)):
    for k in ('k_stitch_chains', 'k_pair_chains', 'k_project_chains', 'k_dp_items'):
        if k in r['Kernel_Name']:
            d.setdefault(k, []).append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6)
try:
    j = json.loads(open(sys.argv[3]).read().strip().splitlines()[-1]); ms = j['ms_per_step']
except Exception as e:
    ms = None
print("== by_row draw", sys.argv[2], "resident ms/step (under the tracer)", ms, {k: [round(x, 2) for x in sorted(v)] for k, v in d.items()})
PY
  rm -rf $R/gpurun_out/prof_st
done
