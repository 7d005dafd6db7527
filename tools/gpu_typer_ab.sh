#!/bin/bash
# A/B of the rows-per-block tiling of k_pair_loglik on one box
set -u
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd "$GRAFT_REPO_ROOT"
for rows in 1 2 4; do
  rm -rf /tmp/ab$rows && mkdir -p /tmp/ab$rows && cp -r hla-la_amd include tools tests __graft_entry__.py /tmp/ab$rows/
  ( cd /tmp/ab$rows && touch hla-la_amd/csrc/hlala_api.hip && make -s -C hla-la_amd/csrc ../libhlala_gpu.so EXTRA=-DHLALA_PAIRLL_ROWS=$rows 2>&1 | grep -E "rror" )
  echo "rows=$rows"; ( cd /tmp/ab$rows && cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ab$rows/prof -- python3 /tmp/ab$rows/tools/typer_profile.py 2>&1 | grep "C=" ; grep k_pair_loglik /tmp/ab$rows/prof/*/*kernel_stats.csv | cut -d, -f2-7 )
done
