#!/bin/bash
# plain bench run(s) on the GPU box: tools/gpu_bench.sh <tag> [bench.py arguments]
set -u
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd "$GRAFT_REPO_ROOT"
TAG=${1:-r02}; shift || true
mkdir -p gpurun_out
make -s -C oracle 2>&1 | tail -1
make -s -C tools/graphm 2>&1 | tail -1
nproc; grep -m1 "model name" /proc/cpuinfo
( time timeout 1700 python bench.py "$@" ) > gpurun_out/bench_$TAG.log 2>&1
tail -4 gpurun_out/bench_$TAG.log | cut -c1-6000
