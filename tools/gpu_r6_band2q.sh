#!/bin/bash
# round 6: quick check after a change of the two-track band kernels: parity on the alignment suite + Graph M, then the class statistics of a Graph M batch
set -u
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1800 python -m pytest tests/test_gpu_align.py tests/test_graph_m.py -x -q -m gpu 2>&1 | tail -6 | tee gpurun_out/r6_band2q_tests.log
for v in ${VARIANTS:-1}; do
  echo "== HLALA_DP_BAND2=$v ${AB_ENV:-}"
  env HLALA_DP_BAND2=$v ${AB_ENV:-} timeout 600 python tools/band_stats.py 1048576 5000000 2>&1 | grep -E "band|16-lane|later|stages"
done
if [ -n "${KSTATS:-}" ]; then
  cd /tmp && export TMPDIR=/tmp
  rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_b2
  rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_b2 -- python3 $GRAFT_REPO_ROOT/tools/band_stats.py 1048576 5000000 > /dev/null 2>&1
  python3 - $GRAFT_REPO_ROOT/gpurun_out/prof_b2 <<'PY'
import csv, glob, sys, re
f = glob.glob(sys.argv[1] + '/*/*kernel_stats.csv')[0]
for r in csv.DictReader(open(f)):
    n = r['Name']
    if 'k_dp' in n or 'k_pair' in n or 'k_project' in n or 'k_rethread' in n or 'k_stitch' in n:
        print("%-60s calls %4s avg %9.3f ms" % (re.sub(r'hlala::|\(.*', '', n)[:60], r['Calls'], float(r['AverageNs']) / 1e6))
PY
  rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_b2
fi
