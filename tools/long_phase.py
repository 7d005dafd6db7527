"""Phase clocks of the long-read projection (HLALA_DEBUG=1: k_project_chains<ProjLdsLong> adds its cycles per phase to counters[16..21]): n distinct reads of ~10 kb on Graph M.
   HLALA_DEBUG=1 python tools/long_phase.py [reads] [levels]"""
import os, sys, time, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from tools import synth
from conftest import load_package
P = load_package()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8000
L = int(sys.argv[2]) if len(sys.argv) > 2 else 5000000
w = synth.make_world_m(seed=2, n_levels=L)
bs = synth.make_long_batches_parallel(w, n, per_batch=n, seed=700, len_lo=6000, len_hi=14000, procs=8)
ctx = P.Context(w["graph"], w["contigs"], insert_mean=200.0, insert_sd=35.0, rng_seed=12345, long_read_mode=1, max_columns=16384)
gb = ctx.batch_unpaired(bs[0])
gb.align(); gb.stats()
t = time.perf_counter(); gb.align(); st = gb.stats(); dt = time.perf_counter() - t
print("%d reads (%.1f Mbases): project %.1f ms, pad + score %.1f ms, select %.1f ms -> %.0f reads/s; errors %d" % (n, bs[0]["read_off"][-1] / 1e6, st.ms_project, st.ms_extend, st.ms_pair, n / dt, st.n_errors))
buf = (C.c_ulonglong * 32)()
ctx.lib.hlala_debug_counters.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_ulonglong)]
ctx.lib.hlala_debug_counters(ctx.h, gb.b, buf)
t = np.array(list(buf)[16:24], dtype=np.float64)
if t[7] > 0:
    names = ["CIGAR walk -> columns", "trim / pad", "cleanInitialAlignment", "restrict to no-gap areas", "re-threading DP", "backtrace + outputs"]
    print("cycles per read (%d reads): " % int(t[7]) + "; ".join("%s %.0f k" % (nm, t[i] / t[7] / 1e3) for i, nm in enumerate(names)) + "; sum %.0f k" % (t[:6].sum() / t[7] / 1e3))
