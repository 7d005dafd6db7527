#!/bin/bash
# the in-memory class's LDS sort scratch: parity of the slab fallback (forced with a 512-pair scratch), then timing of 8192 / 4096 / 2048 pairs
set -u
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
rm -rf /tmp/vs && mkdir /tmp/vs && cp -r hla-la_amd include tools tests oracle bench.py __graft_entry__.py /tmp/vs/
( cd /tmp/vs && rm -rf hla-la_amd/csrc/_obj/hlala_api.o && make -s -C hla-la_amd/csrc ../libhlala_gpu.so EXTRA="-DHLALA_DP_SORT_SCRATCH=512" 2>&1 | grep -E "rror"
  timeout 900 python -m pytest tests/test_full_scale.py -m gpu -x -q -k "in_memory" 2>&1 | tail -3 ) | tee gpurun_out/r3_sort_scratch.log
bash tools/gpu_tail_variants.sh "-DHLALA_DP_SORT_SCRATCH=8192" "" "-DHLALA_DP_SORT_SCRATCH=2048" 2>&1 | tee -a gpurun_out/r3_sort_scratch.log
