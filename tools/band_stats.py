"""Class statistics of one Graph M batch: how many DP calls the band kernel took / passed on, the time of each class (one batch alone on the device).
   python tools/band_stats.py <pairs> <levels> [frac_gene]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from tools import synth
from conftest import load_package
P = load_package()
n_pairs = int(sys.argv[1]); G = int(sys.argv[2]); fg = float(sys.argv[3]) if len(sys.argv) > 3 else 0.3
w = synth.make_world_m(seed=2, n_levels=G)
b = synth.make_batch_m(w, n_pairs, seed=1000, frac_gene=fg)
ctx = P.Context(w["graph"], w["contigs"], insert_mean=b["insert_mean"], insert_sd=b["insert_sd"], rng_seed=12345)
gb = ctx.batch(b)
gb.align(); gb.stats()
gb.align(); st = gb.stats()
print("pairs %d levels %d frac_gene %.2f: DP calls %d (shared %d), iterations %d, cells %d, errors %d" % (n_pairs, G, fg, st.n_dp_calls, st.n_dp_shared, st.n_dp_iterations, st.n_dp_cells, st.n_errors))
print(" band: %d calls (%.1f %% of the calls that run), failed over %d (%.2f %%), %.2f ms" % (st.n_dp_band, 100.0 * st.n_dp_band / max(1, st.n_dp_band + st.n_dp_class[0] - st.n_dp_band_failed), st.n_dp_band_failed, 100.0 * st.n_dp_band_failed / max(1, st.n_dp_band), st.ms_dp_band))
import ctypes as C
wc = gb.work_counters(); why = P.DEBUG_WC_BAND_WHY
print("  fail-over reasons: past the staged levels %d, past the linear run %d, too many iterations %d, too many ties %d" % (wc[why + 2], wc[why + 3], wc[why + 4], wc[why + 5]))
ls = [wc[P.DEBUG_WC_BAND_FETCH + k] for k in range(6)]
print("  items fetched (>= listed) by the 16 / 32 / 64-lane band kernels, left + right:", ls[0] + ls[1], ls[2] + ls[3], ls[4] + ls[5])
why2 = P.DEBUG_WC_N - 8
print(" two-track band: %d calls (%.1f %% of the calls that run), failed over %d (%.2f %%): end of the track steps %d, iterations %d, ties %d, other %d; %.2f ms" % (st.n_dp_band2, 100.0 * st.n_dp_band2 / max(1, st.n_dp_band + st.n_dp_band2 + st.n_dp_class[0] - st.n_dp_band_failed - st.n_dp_band2_failed),
      st.n_dp_band2_failed, 100.0 * st.n_dp_band2_failed / max(1, st.n_dp_band2), wc[why2 + 2], wc[why2 + 4], wc[why2 + 5], wc[why2 + 6] + wc[why2 + 7], st.ms_dp_band2))
print(" 16-lane class: %d calls, %.2f ms (jump-free part: %d calls, %d met a jump, %.2f ms)" % (st.n_dp_class[0], st.ms_dp_class[0], st.n_dp_jump_free, st.n_dp_jump_free_failed, st.ms_dp_jump_free))
print(" later classes: calls", list(st.n_dp_class)[1:], "ms", [round(x, 2) for x in list(st.ms_dp_class)[1:]])
print(" stages ms: project %.2f extend %.2f pair %.2f -> %.0f pairs/s (one batch alone)" % (st.ms_project, st.ms_extend, st.ms_pair, n_pairs / ((st.ms_project + st.ms_extend + st.ms_pair) * 1e-3)))
buf = (C.c_ulonglong * 32)()
ctx.lib.hlala_debug_counters.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_ulonglong)]
ctx.lib.hlala_debug_counters(ctx.h, gb.b, buf)
for nm, o in (("16", 8), ("32", 16), ("64", 24)):
    t2 = np.array(list(buf)[o:o + 7], dtype=np.float64)
    if os.environ.get("B2_TIMING") and t2[6] > 0:
        print(" two-track band kernel (%s lanes per call), cycles per task of a wavefront (-DHLALA_B2_TIMING): draw + stage %.0f, iterations %.0f (%.1f per task, %.0f cycles each), end cell + backtrace %.0f, columns + outputs %.0f, between tasks %.0f | %d tasks" % (nm, t2[0] / t2[6], t2[1] / t2[6], t2[5] / t2[6], t2[1] / max(1, t2[5]), t2[2] / t2[6], t2[3] / t2[6], t2[4] / t2[6], int(t2[6])))
t = np.array(list(buf)[16:23], dtype=np.float64)
if t[6] > 0:
    print(" band kernel (16 lanes), cycles per task of a wavefront (-DHLALA_BAND_TIMING): draw + stage %.0f, iterations %.0f (%.1f per task, %.0f cycles each), end cell + backtrace %.0f, columns + outputs %.0f, between tasks %.0f | %d tasks (all three kernels)" % (t[0] / t[6], t[1] / t[6], t[5] / t[6], t[1] / max(1, t[5]), t[2] / t[6], t[3] / t[6], t[4] / t[6], int(t[6])))
