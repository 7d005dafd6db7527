#!/bin/bash
# round 5: the whole GPU suite, then the bench line as the driver runs it
set -u
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( time timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -12 ) > gpurun_out/r5_full_tests.log 2>&1
tail -14 gpurun_out/r5_full_tests.log
timeout 1200 python bench.py > gpurun_out/r5_bench.json 2> gpurun_out/r5_bench.err
tail -1 gpurun_out/r5_bench.json | cut -c1-700
