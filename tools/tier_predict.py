"""How well does the widest level within a DP call's reach predict the capacity class the call ends in?  One Graph M batch through the product, then the items and the
retry lists of every class (hlala_debug_dp_items): for each class k the distribution of `widest level within reach` of the calls that entered it.
   python tools/tier_predict.py <pairs> <levels> [frac_gene]"""
import sys, os, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from tools import synth
from conftest import load_package
P = load_package()
n_pairs = int(sys.argv[1]); G = int(sys.argv[2]); fg = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
w = synth.make_world_m(seed=2, n_levels=G)
b = synth.make_batch_m(w, n_pairs, seed=77, frac_gene=fg)
ctx = P.Context(w["graph"], w["contigs"], insert_mean=b["insert_mean"], insert_sd=b["insert_sd"], rng_seed=12345)
gb = ctx.batch(b); gb.align(); st = gb.stats()
nc = b["n_chains"]
items, retry = gb.dp_items(); wc = gb.work_counters()
npl = np.bincount(w["graph"]["node_level"], minlength=w["graph"]["n_levels"]).astype(np.int64)
L = len(npl)
# widest level within R levels ahead / behind: sliding maximum
R = 64
def slide_max(a, R):
    out = a.copy()
    sh = 1
    cur = a.copy()
    while sh < R:
        cur = np.maximum(cur, np.concatenate([cur[sh:], np.zeros(sh, cur.dtype)])); sh *= 2
    return cur      # max over [l, l + 2^k) with 2^k >= R
wf = slide_max(npl, R); wb = slide_max(npl[::-1], R)[::-1]
valid = items[:, 0] >= 0
slot = np.arange(2 * nc); isR = slot >= nc
lvl = items[:, 4]
wmax = np.where(isR, wf[np.clip(lvl, 0, L - 1)], wb[np.clip(lvl, 0, L - 1)])
print("calls", int(valid.sum()), "by class entered:", list(st.n_dp_class), "band", st.n_dp_band)
qs = [50, 90, 99, 100]
print("all calls: widest level within %d levels, percentiles %s: %s" % (R, qs, np.percentile(wmax[valid], qs)))
ent = {}
for k in range(1, 7):
    sl = []
    for p in range(2):
        cnt = wc[12 + 4 * (k - 1) + 2 * p]
        sl.append(retry[(2 * (k - 1) + p) * nc:(2 * (k - 1) + p) * nc + cnt])
    sl = np.concatenate(sl); ent[k] = sl
    if len(sl):
        print("class %d: %6d calls entered; widest level: min %d, percentiles 1 / 10 / 50 / 90: %s" % (k, len(sl), wmax[sl].min(), np.percentile(wmax[sl], [1, 10, 50, 90])))
for k in (4, 5, 6):
    if len(ent[k]) == 0: continue
    for thr in (100, 150, 200, 250, 300, 350, 400):
        caught = int((wmax[ent[k]] >= thr).sum()); flagged = int((wmax[valid] >= thr).sum())
        print("  predictor `widest >= %d`: catches %d of %d class-%d calls, flags %d calls in all" % (thr, caught, len(ent[k]), k, flagged))
