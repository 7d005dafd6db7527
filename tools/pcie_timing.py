"""Host-buffer boundary cost: hlala_batch_create (upload) + hlala_align_batch + hlala_batch_get_pairs (download) for one batch."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from tools import synth
from conftest import load_package
P = load_package()
n_pairs = int(sys.argv[1]); G = int(sys.argv[2])
w = synth.make_world(seed=2, G=G, k=1, n_mut=3)
b = synth.make_batch_fast(w, n_pairs, seed=1000)
ctx = P.Context(w["graph"], w["contigs"], insert_mean=b["insert_mean"], insert_sd=b["insert_sd"], rng_seed=12345)
for it in range(3):
    t0 = time.perf_counter(); gb = ctx.batch(b); t1 = time.perf_counter(); gb.align(); st = gb.stats(); t2 = time.perf_counter(); pr = gb.pairs(); t3 = time.perf_counter()
    up = sum(v.nbytes for v in b.values() if hasattr(v, 'nbytes')); dn = sum(v.nbytes for v in pr.values() if hasattr(v, 'nbytes'))
    print('run %d: create+upload %.1f ms (%.2f GB), align %.1f ms, get_pairs %.1f ms (%.2f GB) -> %.0f pairs/s host-buffer inclusive' % (it, (t1-t0)*1e3, up/1e9, (t2-t1)*1e3, (t3-t2)*1e3, dn/1e9, n_pairs/(t3-t0)))
    del gb
for it in range(3):      # the same with the packed download (columns without padding) + the per-pair scalars
    t0 = time.perf_counter(); gb = ctx.batch(b); t1 = time.perf_counter(); gb.align(); st = gb.stats(); t2 = time.perf_counter(); pk = gb.pairs_packed(); t3 = time.perf_counter()
    dn = sum(v.nbytes for v in pk.values() if hasattr(v, 'nbytes'))
    print('packed %d: create+upload %.1f ms, align %.1f ms, get_pairs_packed %.1f ms (%.2f GB) -> %.0f pairs/s host-buffer inclusive' % (it, (t1-t0)*1e3, (t2-t1)*1e3, (t3-t2)*1e3, dn/1e9, n_pairs/(t3-t0)))
    del gb
