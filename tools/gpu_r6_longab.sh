#!/bin/bash
# round 6: the long-read projection, this build against a base build kept under gpurun_in_ab/base.so (HLALA_LIB_PATH), alternating on ONE box: 50 000 distinct reads in one batch
set -u
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd $GRAFT_REPO_ROOT
for i in 1 2; do
  echo "base"; HLALA_LIB_PATH=$GRAFT_REPO_ROOT/gpurun_in_ab/base.so timeout 900 python tools/long_phase.py 50000 5000000 2>&1 | grep "reads/s"
  echo "new";  timeout 900 python tools/long_phase.py 50000 5000000 2>&1 | grep "reads/s"
done
[ "${1:-}" = "debug" ] && HLALA_DEBUG=1 timeout 900 python tools/long_phase.py 50000 5000000 2>&1 | tail -12
[ "${1:-}" = "tests" ] && timeout 1800 python -m pytest tests/test_long_reads_full.py tests/test_unpaired.py -x -q -m gpu 2>&1 | tail -3
exit 0
