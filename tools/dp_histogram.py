"""Per-DP-call widest frontier / largest target set of the Graph M workload, from the CPU oracle (capacity classes of kernel_dp.hip):
   python tools/dp_histogram.py"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from tools import synth
from oracle_binding import Oracle
w = synth.make_world_m(seed=2, n_levels=1_000_000)
for fg in (0.0, 1.0):
    b = synth.make_batch_m(w, 3000, seed=1000, frac_gene=fg)
    o = Oracle(w["graph"], w["contigs"], insert_mean=b["insert_mean"], insert_sd=b["insert_sd"], rng_seed=12345, max_columns=384)
    o.dp_histogram(reset=True)
    o.align_batch(b)
    h = o.dp_histogram()
    print("frac_gene", fg, "calls", h["frontier"][:15].sum())
    print(" frontier buckets (<=1,2,4,8,16,32,64,...):", (h["frontier"][:15]/h["frontier"][:15].sum()).round(3)[:10])
    print(" targets  buckets:", (h["targets"][:14]/h["targets"][:14].sum()).round(3)[:10])
    # (the last buckets are never reached; the oracle reuses them for the calls that outgrow the 16-lane class: count, sum of the iteration of
    # the first overflow, sum of their iterations)
    n = max(1, int(h["frontier"][15]))
    print(" calls that pass 16 frontier cells or 24 targets: %d, first at iteration %.1f of %.1f on average" % (h["frontier"][15], h["targets"][14] / n, h["targets"][15] / n))
