#!/bin/bash
# tools/gpu_pytest.sh <pytest arguments>: GPU tests on the box
set -u
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
make -s -C oracle 2>&1 | tail -1
make -s -C tools/graphm 2>&1 | tail -1
( time timeout 2400 python -m pytest "$@" ) > gpurun_out/pytest.log 2>&1
tail -40 gpurun_out/pytest.log
