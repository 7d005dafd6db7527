#!/bin/bash
# round 6: the long-read projection after the LDS-only synchronisation of its level loops and the hybrid form: parity (long-read tests, unpaired, the alignment suite's
# projection), then the phase clocks and the 50 000-read batch
set -u
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd $GRAFT_REPO_ROOT
timeout 1800 python -m pytest tests/test_long_reads_full.py tests/test_unpaired.py tests/test_gpu_align.py tests/test_graph_m.py -x -q -m gpu 2>&1 | tail -6
HLALA_DEBUG=1 timeout 900 python tools/long_phase.py 8000 5000000 2>&1 | tail -2
timeout 900 python tools/long_phase.py 50000 5000000 2>&1 | tail -1
