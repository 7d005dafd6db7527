"""Timings of the typer-side entry points on a large resident batch (host wall clock incl. transfers of their outputs)."""
import os, sys, time
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
gb = ctx.batch(b); gb.align()
def T(f, *a, **k):
    t = time.perf_counter(); r = f(*a, **k); return r, (time.perf_counter() - t) * 1e3
# 17 "genes" of 6 kb spread over the graph; one locus = two exons of 270 / 276 levels inside the first gene
starts = np.linspace(G // 20, G - G // 20, 17).astype(np.int32); ctx.set_gene_intervals(starts, starts + 6000)
inc, t_post = T(gb.postprocess)
lmin = int(starts[8]) + 500; l2e = np.full(1300, -1, np.int32); l2e[:270] = np.arange(270); l2e[900:1176] = np.arange(270, 546)
e, t_pos = T(gb.exon_positions, lmin, l2e, b["insert_mean"], b["insert_sd"], pair_mask=inc)
us, t_us = T(gb.unit_stats)
q = ["".join(np.random.default_rng(i).choice(list("ACGT"), 31)) for i in range(1000)]
pr, t_km = T(ctx.kmer_presence, gb, q, 31, inc)
print("%d pairs: postprocess %.1f ms (%d pairs overlap a gene); exon positions of one locus %.1f ms (%d reads, %d positions); unit stats %.1f ms; k-mer presence (1000 queries) %.1f ms"
      % (n_pairs, t_post, int(inc.sum()), t_pos, e["n_reads"], e["n_pos"], t_us, t_km))
