#!/bin/bash
# round 6: long reads handed out heaviest window first (batch.h: order_cost) against position order, 50 000 distinct reads in one batch; then the long-read parity tests
set -u
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd $GRAFT_REPO_ROOT
for o in 0 1 0 1; do echo "HLALA_LONG_ORDER=$o"; HLALA_LONG_ORDER=$o timeout 900 python tools/long_phase.py 50000 5000000 2>&1 | tail -1; done
timeout 1800 python -m pytest tests/test_long_reads_full.py tests/test_unpaired.py -x -q -m gpu 2>&1 | tail -3
