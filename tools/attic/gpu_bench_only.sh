#!/bin/bash
# the default bench line alone (after `python tools/derive_profiles.py <tag>`, so that the line carries the traffic of the current kernels)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
( time timeout 1500 python bench.py ) > gpurun_out/r3_bench_full.log 2> gpurun_out/r3_bench_full.err
tail -3 gpurun_out/r3_bench_full.err
grep '^{' gpurun_out/r3_bench_full.log | tail -1 | cut -c1-1500
