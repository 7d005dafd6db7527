"""Slowest DP calls of one capacity class (build: make EXTRA=-DHLALA_DP_PROFILE=<tier>; run with HLALA_DEBUG=1)."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge  # noqa: E402
from tools import synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
fg = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
levels = int(sys.argv[3]) if len(sys.argv) > 3 else 5_000_000
P = ge.load_package()
w = synth.make_world_m(seed=2, n_levels=levels)
b = synth.make_batch_m(w, n, seed=77, frac_gene=fg)
ctx = P.Context(w["graph"], w["contigs"], insert_mean=b["insert_mean"], insert_sd=b["insert_sd"], rng_seed=12345)
gb = ctx.batch(b)
gb.align(); st = gb.stats()
buf = np.zeros(8192, np.int32)
ctx.lib.hlala_debug_buffer.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
ctx.lib.hlala_debug_buffer(ctx.h, None, 1)
gb.align(); st = gb.stats()
rc = ctx.lib.hlala_debug_buffer(ctx.h, buf.ctypes.data, 1)
print("rc", rc, "class ms", [round(float(x), 1) for x in st.ms_dp_class], "calls", list(st.n_dp_class), "errors", st.n_errors)
nrec = int(buf[0]); print("records over the threshold:", nrec)
r = buf[16:16 + 16 * min(nrec, 500)].reshape(-1, 16)
r = r[np.argsort(-r[:, 7])]
print("item iters cellsEval nCells slowIters sumImp preIters kcycles maxNT | kcycles in: records pushes tlist early-lookups evaluate post-evaluate filter+sort+reset")
for x in r[:40]:
    print(" ".join(str(int(v)) for v in x))
print("sum kcycles", int(r[:, 7].sum()), "mean", float(r[:, 7].mean()) if len(r) else 0)

# groups: by whether the call looked up early cells (preIters > 0) and whether it met one again (slowIters > 0)
if len(r):
    pre = r[:, 6] > 0; slow = r[:, 4] > 0
    for name, m in (("no early cell", ~pre), ("early-cell look-ups, none met again", pre & ~slow), ("met an early cell again (two-pass iterations)", slow)):
        if m.any():
            x = r[m]
            print("%-48s %4d calls: mean kcycles %7.1f, iterations %5.1f, cells %6.1f, slow iters %4.1f, pre iters %4.1f | records %5.1f pushes %5.1f tlist %4.1f early %5.1f evaluate %5.1f post %5.1f filter %5.1f" % (
                name, int(m.sum()), x[:, 7].mean(), x[:, 1].mean(), x[:, 2].mean(), x[:, 4].mean(), x[:, 6].mean(), x[:, 9].mean(), x[:, 10].mean(), x[:, 11].mean(), x[:, 12].mean(), x[:, 13].mean(), x[:, 14].mean(), x[:, 15].mean()))
