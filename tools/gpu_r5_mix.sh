#!/bin/bash
# round 5: where the extension stage's time goes by kind of pair: backbone only / gene windows only / the bench's mix, 1 M pairs each (one batch alone)
set -u
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd $GRAFT_REPO_ROOT
for fg in 0.0 1.0 0.3; do
  echo "== frac_gene $fg"
  timeout 600 python tools/band_stats.py ${MIX_PAIRS:-1048576} 5000000 $fg 2>&1 | grep -E "pairs|band:|16-lane|later|stages"
done
