import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from tools import synth
from conftest import load_package
from oracle_binding import Oracle
P = load_package()
n = 6000; seed = 102
w = synth.make_world(seed=seed, G=20000, k=0, n_largegap=2)
b = synth.make_batch(w, n, seed=seed + 1000, p_secondary=0.8, max_secondary=5)
kwc = dict(insert_mean=b["insert_mean"], insert_sd=b["insert_sd"], rng_seed=seed, max_columns=384)
exp = Oracle(w["graph"], w["contigs"], **kwc).align_batch(b)["pairs"]
ctx = P.Context(w["graph"], w["contigs"], **kwc)
gb = ctx.batch(b); gb.align(); got = gb.pairs()
g = got["col_mapq"].reshape(2 * n, 384); e = exp["col_mapq"].reshape(2 * n, 384)
bad = np.nonzero((g != e).any(1))[0]
print("reads with differing col_mapq:", len(bad), bad[:20])
for r in bad[:6]:
    p = r // 2; nc = got["n_cols"][r]
    d = np.nonzero(g[r] != e[r])[0]
    print("read", r, "pair", p, "n_comb", got["n_combinations"][p], exp["n_combinations"][p], "ncols", nc, "diff cols", len(d), d[:10], "got", g[r][d[:10]], "exp", e[r][d[:10]],
          "pair_mapq", got["pair_mapq"][p], exp["pair_mapq"][p], "mate_mapq", got["mate_mapq"][r], exp["mate_mapq"][r], "chains of read", b["chain_off"][r + 1] - b["chain_off"][r])
