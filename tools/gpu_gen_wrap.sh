#!/bin/bash
# parity tests on a build whose rare paths are the common ones: early-cell table generations wrap around after a handful of DP calls per slab
# (kernel_dp.hip: DP_EARLY_GEN_MAX), pairs with more than two chain combinations keep their table in the HBM scratch (kernel_pair.hip: PAIR_COMB_LDS)
set -u
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd "$GRAFT_REPO_ROOT"
make -s -C oracle 2>&1 | tail -1
make -s -C tools/graphm 2>&1 | tail -1
rm -rf /tmp/vw && mkdir /tmp/vw && cp -r hla-la_amd include tools tests oracle __graft_entry__.py /tmp/vw/
( cd /tmp/vw && touch hla-la_amd/csrc/hlala_api.hip && make -s -C hla-la_amd/csrc ../libhlala_gpu.so EXTRA="-DHLALA_EARLY_GEN_MAX=3 -DHLALA_PAIR_COMB_LDS=2" 2>&1 | grep -E "rror" )
( cd /tmp/vw && timeout 1200 python -m pytest tests/test_graph_m.py tests/test_gpu_extend.py tests/test_gpu_align.py tests/test_unpaired.py -x -q -m gpu 2>&1 | tail -3 )
