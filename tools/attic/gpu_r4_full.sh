#!/bin/bash
# round 4: the whole GPU suite, then the default bench line (what the driver runs), then a 2-rank dry run of the N > 1 path over gloo on the one GPU
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
make -s -C oracle 2>&1 | tail -1; make -s -C tools/graphm 2>&1 | tail -1
( time timeout 2400 python -m pytest tests -m gpu -x -q ) > gpurun_out/r4_pytest_full.log 2>&1
tail -6 gpurun_out/r4_pytest_full.log
( time timeout 1500 python bench.py --steps 10 --warmup 3 ) > gpurun_out/r4_bench_full.log 2> gpurun_out/r4_bench_full.err
tail -c 6000 gpurun_out/r4_bench_full.log | cut -c1-3000; tail -5 gpurun_out/r4_bench_full.err
