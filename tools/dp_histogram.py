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
    print("frac_gene", fg, "calls", h["frontier"].sum())
    print(" frontier buckets (<=1,2,4,8,16,32,64,...):", (h["frontier"]/h["frontier"].sum()).round(3)[:10])
    print(" targets  buckets:", (h["targets"]/h["targets"].sum()).round(3)[:10])
