#!/bin/bash
# round 5: per-kernel times of one Graph M batch (rocprofv3 --kernel-trace --stats over tools/band_stats.py)
set -u
: "${GRAFT_REPO_ROOT:?run through gpurun}"
R=$GRAFT_REPO_ROOT
PAIRS=${1:-262144}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_band
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_band -- python3 $R/tools/band_stats.py $PAIRS 5000000 > $R/gpurun_out/prof_band.log 2>&1
find $R/gpurun_out/prof_band -name "*kernel_trace.csv" -delete
cat $R/gpurun_out/prof_band/*/*kernel_stats.csv | cut -c1-160 | head -32
tail -9 $R/gpurun_out/prof_band.log
