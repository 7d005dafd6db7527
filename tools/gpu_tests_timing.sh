#!/bin/bash
set -u
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd $GRAFT_REPO_ROOT
make -C oracle 2>&1 | tail -1
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -5
timeout 300 python tools/dbg_timing.py 262144 5000000 2>&1 | tail -3
