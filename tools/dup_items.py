"""How many extension DPs of a batch start from the same cell of the same read (same mate, direction, read offset, graph node)?"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from tools import synth
from conftest import load_package
P = load_package()
n_pairs = int(sys.argv[1]); G = int(sys.argv[2])
w = synth.make_world(seed=2, G=G, k=1, n_mut=3)
b = synth.make_batch_fast(w, n_pairs, seed=1000)
ctx = P.Context(w["graph"], w["contigs"], insert_mean=b["insert_mean"], insert_sd=b["insert_sd"], rng_seed=12345)
gb = ctx.batch(b); gb.project()
s = gb.chains(0)
nc = b["n_chains"]; stride = s["_stride"]
read_of = np.repeat(np.arange(2 * n_pairs), np.diff(b["chain_off"]))
rlen = np.diff(b["read_off"])[read_of]
ok = s["status"] == 0
e = s["col_edge"].reshape(nc, stride)
first_e = e[:, 0]; last_e = e[np.arange(nc), np.maximum(s["n_cols"] - 1, 0)]
g = w["graph"]
needL = ok & (s["seq_begin"] != 0); needR = ok & (s["seq_end"] != rlen - 1)
keyL = np.stack([read_of, s["seq_begin"], g["edge_from"][np.where(needL, first_e, 0)]], 1)[needL]
keyR = np.stack([read_of, s["seq_end"], g["edge_to"][np.where(needR, last_e, 0)]], 1)[needR]
for nm, k in (("left", keyL), ("right", keyR)):
    u = np.unique(k, axis=0)
    print(nm, "items", len(k), "distinct start cells", len(u), "-> duplicates %.1f %%" % (100 * (1 - len(u) / max(1, len(k)))))
