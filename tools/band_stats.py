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
print(" 16-lane class: %d calls, %.2f ms (jump-free part: %d calls, %d met a jump, %.2f ms)" % (st.n_dp_class[0], st.ms_dp_class[0], st.n_dp_jump_free, st.n_dp_jump_free_failed, st.ms_dp_jump_free))
print(" later classes: calls", list(st.n_dp_class)[1:], "ms", [round(x, 2) for x in list(st.ms_dp_class)[1:]])
print(" stages ms: project %.2f extend %.2f pair %.2f -> %.0f pairs/s (one batch alone)" % (st.ms_project, st.ms_extend, st.ms_pair, n_pairs / ((st.ms_project + st.ms_extend + st.ms_pair) * 1e-3)))
