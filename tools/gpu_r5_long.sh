#!/bin/bash
# round 5: config 5 (long reads) alone -- rocprofv3 kernel stats, then the SQ and the HBM counter passes of the projection kernel (each in its own run)
set -u
: "${GRAFT_REPO_ROOT:?run through gpurun}"
R=$GRAFT_REPO_ROOT
TAG=${1:-r05_long}
N=${2:-50000}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/${TAG}_stats $R/gpurun_out/${TAG}_sq $R/gpurun_out/${TAG}_fetch $R/gpurun_out/${TAG}_write
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_stats -- python3 $R/tools/long_profile.py $N > $R/gpurun_out/${TAG}_stats.log 2>&1
grep "long reads:" $R/gpurun_out/${TAG}_stats.log
if [ "${LONG_PMC:-1}" = 1 ]; then
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_sq -- python3 $R/tools/long_profile.py 20000 > $R/gpurun_out/${TAG}_sq.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_fetch -- python3 $R/tools/long_profile.py 20000 > $R/gpurun_out/${TAG}_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_write -- python3 $R/tools/long_profile.py 20000 > $R/gpurun_out/${TAG}_write.log 2>&1
fi
find $R/gpurun_out/${TAG}_stats $R/gpurun_out/${TAG}_sq $R/gpurun_out/${TAG}_fetch $R/gpurun_out/${TAG}_write -name "*kernel_trace.csv" -delete 2>/dev/null
python3 - $R/gpurun_out $TAG <<'PY'
import csv, glob, sys, collections
d, tag = sys.argv[1], sys.argv[2]
for f in glob.glob(f"{d}/{tag}_stats/*/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if float(r["TotalDurationNs"]) > 2e6: print("%-70s calls %3s total %9.2f ms avg %9.2f ms" % (r["Name"][:70], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e6))
for sub in ("sq", "fetch", "write"):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
    for f in glob.glob(f"{d}/{tag}_{sub}/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:60]; acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
    for k in acc:
        if "project" in k or "stitch" in k: print(sub, k, len(n[k]), "launches:", {c: "%.4g" % (v / len(n[k])) for c, v in acc[k].items()})
PY
