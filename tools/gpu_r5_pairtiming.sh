#!/bin/bash
# round 5: phase clocks of k_pair_chains (timing build): cycles per wavefront summed over the grid -> share of each phase
set -u
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd $GRAFT_REPO_ROOT
touch hla-la_amd/csrc/kernel_pair.hip
make -C hla-la_amd/csrc EXTRA="-DHLALA_PAIR_TIMING" 2>&1 | grep -E "error" | head
for fg in 0.3 0.0 1.0; do
timeout 600 python - $fg <<'PY'
import sys, os
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, ctypes as C
from tools import synth
from conftest import load_package
P = load_package()
fg = float(sys.argv[1])
w = synth.make_world_m(seed=2, n_levels=5000000)
b = synth.make_batch_m(w, 262144, seed=1000, frac_gene=fg)
ctx = P.Context(w["graph"], w["contigs"], insert_mean=b["insert_mean"], insert_sd=b["insert_sd"], rng_seed=12345)
gb = ctx.batch(b); gb.align(); gb.stats(); gb.align(); st = gb.stats()
buf = (C.c_ulonglong * 32)()
ctx.lib.hlala_debug_counters.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_ulonglong)]
ctx.lib.hlala_debug_counters(ctx.h, gb.b, buf)
t = np.array(list(buf)[24:31], dtype=np.float64)
tot = t[:5].sum()
print("frac_gene %.1f: pair stage %.2f ms; cycles: lists + draw %.1f %%, combinations %.1f %%, maximum + posterior %.1f %%, per-position (one combination) %.1f %%, per-position (several) %.1f %%; pairs with one combination %d, with several %d; cycles per pair: one %.0f, several: combos %.0f + positions %.0f" % (
    fg, st.ms_pair, 100 * t[0] / tot, 100 * t[1] / tot, 100 * t[2] / tot, 100 * t[3] / tot, 100 * t[4] / tot, t[5], t[6], (t[3]) / max(1, t[5]), t[1] / max(1, t[5] + t[6]), t[4] / max(1, t[6])))
PY
done
