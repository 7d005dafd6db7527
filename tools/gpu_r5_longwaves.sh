#!/bin/bash
# round 5: config 5, waves of the long-read projection per CU x reads per batch
set -u
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd $GRAFT_REPO_ROOT
for cfg in "12 25000" "16 25000" "12 50000" "16 50000"; do
  set -- $cfg
  echo "== HLALA_PROJ_LONG_WAVES=$1 reads per batch $2"
  LONG_BATCH=$2 HLALA_PROJ_LONG_WAVES=$1 timeout 600 python tools/long_profile.py 50000 5000000 2>&1 | grep "long reads:"
done
