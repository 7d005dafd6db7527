#!/bin/bash
# round 5: k_pair_chains compiled for 5 / 4 / 3 waves per SIMD (96 / 128 / 168 registers: 63 spilled registers at 5): the stage alone and in the fused step
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
for x in 5 4 3; do
  touch hla-la_amd/csrc/kernel_pair.hip
  make -C hla-la_amd/csrc EXTRA="-DHLALA_PAIR_WAVES_PER_SIMD=$x" 2>&1 | grep -E " error" | head
  echo "== build -DHLALA_PAIR_WAVES_PER_SIMD=$x"; run
done
