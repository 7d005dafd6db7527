#!/bin/bash
# round 4, third session: rethread draws of 16, side stream on a CU mask (A/B), HBM traffic of the kernels in position order
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
bash tools/gpu_r4_ab.sh --parity "default:HLALA_X=0" "sidecus64:HLALA_SIDE_CUS=64" "sidecus128:HLALA_SIDE_CUS=128"
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
B="--no-cpu-baseline --resident-only"
rm -rf $R/gpurun_out/r04q_fetch $R/gpurun_out/r04q_write
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/r04q_fetch -- python3 $R/bench.py --steps 1 --warmup 0 --single-batch $B > $R/gpurun_out/r04q_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/r04q_write -- python3 $R/bench.py --steps 1 --warmup 0 --single-batch $B > $R/gpurun_out/r04q_write.log 2>&1
find $R/gpurun_out/r04q_fetch $R/gpurun_out/r04q_write -name "*kernel_trace.csv" -size +8M -delete
cd $R && python3 - <<'PY'
import collections, csv, glob
def counters(d, cn):
    acc = collections.defaultdict(float); n = collections.defaultdict(set)
    for f in glob.glob("gpurun_out/%s/*/*counter_collection.csv" % d):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == cn:
                k = r["Kernel_Name"].split("(")[0][-44:]
                acc[k] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
    return {k: acc[k] / len(n[k]) for k in acc}
f = counters("r04q_fetch", "FETCH_SIZE"); w = counters("r04q_write", "WRITE_SIZE")
for k in sorted(f, key=lambda k: -(2 * f[k] + w.get(k, 0))):
    print("%-46s fetch(x2) %7.2f GB  write %7.2f GB  total %7.2f GB" % (k, 2 * f[k] * 1024 / 1e9, w.get(k, 0) * 1024 / 1e9, (2 * f[k] + w.get(k, 0)) * 1024 / 1e9))
PY
