#!/bin/bash
# round 5: k_stitch_chains: chains per draw -- every launch's duration from the kernel trace of the bench's resident loop (1 M pairs, two batches in flight)
set -u
: "${GRAFT_REPO_ROOT:?run through gpurun}"
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for cfg in ${STITCH_CFGS:-8 12 16}; do
  rm -rf $R/gpurun_out/prof_st
  HLALA_STITCH_DRAW=$cfg rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_st -- python3 $R/bench.py --steps 4 --warmup 1 --resident-only --no-cpu-baseline --long-reads 0 --no-extras > $R/gpurun_out/prof_st.log 2>&1
  python3 - $R/gpurun_out/prof_st $cfg <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/*/*kernel_trace.csv')[0]
d = []
for r in csv.DictReader(open(f)):
    if 'k_stitch_chains' in r['Kernel_Name']:
        d.append((int(r['Start_Timestamp']), (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6, r.get('Grid_Size', r.get('Grid_Size_X', '?'))))
d.sort()
print("== draw", sys.argv[2], "k_stitch_chains launches (ms, grid):", [(round(x[1], 2), x[2]) for x in d])
PY
  rm -rf $R/gpurun_out/prof_st
done
