"""One world of the parity sweep with the differences of the pair records spelled out: python tools/debug_sweep_seed.py <seed> [pairs]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from tools import parity_sweep as ps
from conftest import load_package

cap = {}
def cmp_chains(got, exp, n, **kw):
    cap["ext_got"] = got; cap["ext_exp"] = exp
    return ps_compare(got, exp, n, **kw)
def pairs_equal(got, ep):
    bad = np.nonzero(got["best_chain"] != ep["best_chain"])[0]
    print("reads whose best chain differs:", len(bad), bad[:20])
    for k in ("pair_status", "n_combinations", "strands_valid", "n_cols"):
        print(" ", k, "differs at", np.nonzero(got[k] != ep[k])[0][:10])
    eg = cap["ext_got"]; ee = cap["ext_exp"]
    for r in bad[:6]:
        p = r // 2
        print(" pair", p, "read", r, "best got/exp", got["best_chain"][2 * p:2 * p + 2], ep["best_chain"][2 * p:2 * p + 2], "nComb", got["n_combinations"][p], ep["n_combinations"][p],
              "pair_ll %.17g / %.17g" % (got["pair_ll"][p], ep["pair_ll"][p]), "mapq %.6g / %.6g" % (got["pair_mapq"][p], ep["pair_mapq"][p]))
        for c in sorted(set(list(got["best_chain"][2 * p:2 * p + 2]) + list(ep["best_chain"][2 * p:2 * p + 2]))):
            if c < 0: continue
            n = int(ee["n_cols"][c]); lv = ee["col_level"][c * 384:c * 384 + n]; d = lv[lv >= 0]
            print("   chain", c, "status", ee["status"][c], eg["status"][c], "ncols", n, "ll %.17g / %.17g" % (ee["ll"][c], eg["ll"][c]), "levels", (d[0], d[1], d[-2], d[-1]) if len(d) > 1 else d)
    raise AssertionError("best_chain")
ps_compare = ps.compare_chains
ps.compare_chains = cmp_chains
ps.assert_pairs_equal = pairs_equal
P = load_package()
ps.sweep_world(P, int(sys.argv[1]), int(sys.argv[2]) if len(sys.argv) > 2 else 600)
print("no difference")
