#!/bin/bash
# round 5: the side stream's pairing pass sweeping 64 flags per round: parity, then every launch of k_pair_chains / k_stitch_chains in the kernel trace of the resident loop and the resident step
set -u
: "${GRAFT_REPO_ROOT:?run through gpurun}"
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_graph_m.py -m gpu -x -q > gpurun_out/r5_pairside_pytest.log 2>&1
tail -3 gpurun_out/r5_pairside_pytest.log
for i in 1 2; do
timeout 900 python bench.py --steps 10 --warmup 3 --resident-only --no-cpu-baseline --long-reads 0 --no-extras 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(' resident ms/step %.2f' % d['ms_per_step'], {k: round(v,1) for k,v in d['config']['stage_ms'].items()})"
done
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_ps
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_ps -- python3 $R/bench.py --steps 6 --warmup 2 --resident-only --no-cpu-baseline --long-reads 0 --no-extras > $R/gpurun_out/prof_ps.log 2>&1
python3 - $R/gpurun_out/prof_ps <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/*/*kernel_trace.csv')[0]
d = {}
for r in csv.DictReader(open(f)):
    for k in ('k_pair_chains', 'k_stitch_chains', 'DpWide', 'DpBroad', 'DpLarge', 'DpHuge', 'k_project_chains'):
        if k in r['Kernel_Name']:
            d.setdefault(k, []).append((int(r['Start_Timestamp']), (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6))
for k, v in d.items():
    v.sort(); print(k, [round(x[1], 2) for x in v])
pp = sorted(x[0] for x in d['k_project_chains'])
print("step by the starts of k_project_chains:", [round((b - a) / 1e6, 1) for a, b in zip(pp, pp[1:])])
PY
rm -rf $R/gpurun_out/prof_ps
