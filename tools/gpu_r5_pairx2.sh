#!/bin/bash
# round 5: is k_pair_chains slow beside the side stream's classes?  The three stages called one by one (no side stream: every class on the main stream)
# against the fused entry point; normal build and the early-stop build
set -u
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd $GRAFT_REPO_ROOT
run() { timeout 600 python - <<'PY'
import sys, os
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from tools import synth
from conftest import load_package
P = load_package()
w = synth.make_world_m(seed=2, n_levels=5000000)
b = synth.make_batch_m(w, 1048576, seed=1000, frac_gene=0.3)
ctx = P.Context(w["graph"], w["contigs"], insert_mean=b["insert_mean"], insert_sd=b["insert_sd"], rng_seed=12345)
gb = ctx.batch(b)
for mode in ("fused", "staged", "fused", "staged"):
    if mode == "fused": gb.align()
    else: gb.project(); gb.extend(); gb.pair()
    st = gb.stats()
    print(" %s: project %.2f extend %.2f pair %.2f ms" % (mode, st.ms_project, st.ms_extend, st.ms_pair))
PY
}
for x in "PAIR_X_NONE" "PAIR_X_LEVEL=1"; do
  touch hla-la_amd/csrc/kernel_pair.hip
  make -C hla-la_amd/csrc EXTRA="-D$x" 2>&1 | grep -E " error" | head
  echo "== build -D$x"; run
done
