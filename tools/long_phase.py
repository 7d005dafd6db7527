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
if os.environ.get('HLALA_DEBUG'):
    ctx.lib.hlala_debug_buffer.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.c_int]
    ctx.lib.hlala_debug_buffer(ctx.h, None, 1)
t = time.perf_counter(); gb.align(); st = gb.stats(); dt = time.perf_counter() - t
print("%d reads (%.1f Mbases): project %.1f ms, pad + score %.1f ms, select %.1f ms -> %.0f reads/s; errors %d" % (n, bs[0]["read_off"][-1] / 1e6, st.ms_project, st.ms_extend, st.ms_pair, n / dt, st.n_errors))
buf = (C.c_ulonglong * 32)()
ctx.lib.hlala_debug_counters.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_ulonglong)]
ctx.lib.hlala_debug_counters(ctx.h, gb.b, buf)
t = np.array(list(buf)[16:24], dtype=np.float64)
if t[7] > 0:
    names = ["CIGAR walk -> columns", "trim / pad", "cleanInitialAlignment", "restrict to no-gap areas", "re-threading DP", "backtrace + outputs"]
    print("cycles per read (%d reads): " % int(t[7]) + "; ".join("%s %.0f k" % (nm, t[i] / t[7] / 1e3) for i, nm in enumerate(names)) + "; sum %.0f k" % (t[:6].sum() / t[7] / 1e3))

if os.environ.get('HLALA_DEBUG'):
    dbg = (C.c_int * 8192)()
    ctx.lib.hlala_debug_buffer(ctx.h, dbg, 0)
    cnt = np.array(dbg[4096:4144], dtype=np.int64); sm = np.array(dbg[4160:4208], dtype=np.int64) << 16
    if cnt.sum() > 0:
        print("cycles of a read, by binary logarithm: reads / share of all cycles")
        for k in range(48):
            if cnt[k]: print("  2^%d: %d reads, mean %.2f M cycles, %.1f %% of the cycles" % (k, cnt[k], sm[k] / cnt[k] / 1e6, 100.0 * sm[k] / sm.sum()))
    fine = np.array(dbg[4300:4312], dtype=np.float64) * 4096
    if fine.sum() > 0 and t[7] > 0:
        nm = ["window staging", "level -> column table", "segment list", "short segments, one per lane", "long segments, wave-wide", "chunked / sequential form", "pick reset", "segment backtrace", "outputs", "CIGAR pass 1", "CIGAR pass 2"]
        print("pieces, k cycles per read: " + "; ".join("%s %.0f" % (nm[i], fine[i] / t[7] / 1e3) for i in range(11)))
    hh = np.array(list(buf)[24:28], dtype=np.float64)
    if hh[2] > 0:
        print("level-by-level form: %.0f chunks per read (all reads), %.1f levels per chunk; per chunk %.0f cycles of staging + %.0f of level loops (%.0f per level)" % (hh[2] / t[7], hh[3] / hh[2], hh[0] / hh[2], hh[1] / hh[2], hh[1] / hh[3]))
    nh = dbg[4319]
    if nh > 0:
        ph = np.array(dbg[4320:4326], dtype=np.float64) * 4096 / nh / 1e3; fn = np.array(dbg[4330:4342], dtype=np.float64) * 4096 / nh / 1e3
        print("the %d reads of 2^25 cycles and more, k cycles per read: " % nh + "; ".join("%s %.0f" % (n_, v) for n_, v in zip(names, ph)))
        print("   pieces: " + "; ".join("%s %.0f" % (nm[i], fn[i]) for i in range(9)))
        ch_ = float(dbg[4346]); lv_ = float(dbg[4347])
        if ch_ > 0: print("   level-by-level form: %.0f chunks per read, %.1f levels per chunk; per chunk %.0f cycles of staging + %.0f of level loops (%.0f per level)" % (ch_ / nh, lv_ / ch_, dbg[4344] * 4096.0 / ch_, dbg[4345] * 4096.0 / ch_, dbg[4345] * 4096.0 / lv_))
    nq = min(dbg[4318], 300)
    if nq > 0:
        rec = np.array(dbg[5000:5000 + 10 * nq], dtype=np.int64).reshape(nq, 10)
        print("heavy reads (first %d): chain, CIGAR operations, columns, padded columns, levels, window nodes, segments, long ranges, Mcycles, form (1 par, 2 whole level-by-level, 4 staged, 8 windowed)" % nq)
        o = np.argsort(-rec[:, 8])
        t0 = (rec[:, 9] >> 8).min()
        for r in rec[o][:25]: print("   ", r[0], r[1], r[2], r[3], r[4], r[5], r[6], r[7], "%.1f" % (r[8] * 65536 / 1e6), r[9] & 255, "start at %.1f Mcycles" % (((r[9] >> 8) - t0) * 1.048576))
        rec[:, 9] &= 255
        print("   medians:", np.median(rec[:, 1:8], axis=0), "forms:", np.bincount(rec[:, 9], minlength=16))
    oc = np.array(dbg[4432:4464], dtype=np.float64); osum = np.array(dbg[4400:4432], dtype=np.float64) * 65536; ocol = np.array(dbg[4464:4496], dtype=np.float64) * 16
    print("by the read's ordinal on its wavefront: reads, M cycles per read, cycles per column")
    for k in range(32):
        if oc[k] > 0: print("   %2d: %6d reads, %.1f M cycles, %.0f cycles per column" % (k + 1, oc[k], osum[k] / oc[k] / 1e6, osum[k] / max(ocol[k], 1)))
