"""Regenerates tests/golden/r01_small.npz.

The reference cannot be built or imported in this image (C++ needing Boost + BamTools), so these are NOT
reference outputs: they are the oracle's outputs on a fixed small input, committed as a regression pin for the
oracle itself and as an input/expected-output vector the GPU path is checked against on the GPU box.
Run:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from tools import synth                      # noqa: E402
from oracle_binding import Oracle             # noqa: E402

w = synth.make_world(seed=101, G=3000, k=1)
b = synth.make_batch(w, 40, seed=102)
o = Oracle(w["graph"], w["contigs"], insert_mean=b["insert_mean"], insert_sd=b["insert_sd"], rng_seed=4242, max_columns=384)
r = o.align_batch(b)
out = {}
for k, v in w["graph"].items():
    if k not in ("hap_edge", "hap_node"):
        out["graph__" + k] = v
for k, v in w["contigs"].items():
    out["contigs__" + k] = v
for k, v in b.items():
    out["batch__" + k] = v
st = r["ext"]["_stride"]
ok = r["ext"]["status"] == 0
out["exp__ext_status"] = r["ext"]["status"]; out["exp__ext_ncols"] = r["ext"]["n_cols"]; out["exp__ext_ll"] = r["ext"]["ll"]
out["exp__dp_iters"] = r["ext"]["dp_iters"]; out["exp__dp_score"] = r["ext"]["dp_score"]
# checksums of the chain columns keep the fixture small
lv = r["ext"]["col_level"].reshape(-1, st).astype(np.int64); n = r["ext"]["n_cols"]
mask = np.arange(st)[None, :] < n[:, None]
out["exp__ext_level_sum"] = (lv * mask).sum(1); out["exp__ext_edge_sum"] = (r["ext"]["col_edge"].reshape(-1, st).astype(np.int64) * mask).sum(1)
for k in ("pair_status", "best_chain", "n_combinations", "pair_ll", "pair_mapq", "mate_mapq", "strands_valid", "n_cols", "col_level", "col_gchar", "col_schar", "col_mapq"):
    out["exp__" + k] = r["pairs"][k]
np.savez_compressed(os.path.join(HERE, "r01_small.npz"), **out)
print("wrote", os.path.join(HERE, "r01_small.npz"), "chains", b["n_chains"])
