#!/bin/bash
set -u
: "${GRAFT_REPO_ROOT:?run through gpurun}"
R=$GRAFT_REPO_ROOT
cd "$R"; mkdir -p gpurun_out; rm -rf gpurun_out/r02_typer
timeout 600 python tools/typer_profile.py 2>&1 | tail -3
timeout 900 python -m pytest tests/test_typer.py tests/test_call.py tests/test_typer_chain.py -x -q -m gpu 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r02_typer -- python3 $R/tools/typer_profile.py > $R/gpurun_out/r02_typer.log 2>&1
find $R/gpurun_out/r02_typer -name "*kernel_trace.csv" -size +4M -delete
cat $R/gpurun_out/r02_typer/*/*kernel_stats.csv | cut -c1-160 | head -14
