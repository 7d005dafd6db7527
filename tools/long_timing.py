"""Long-read mode (BASELINE config 5 style) throughput: n reads of ~10 kb, one primary alignment each, max_columns = 16384."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from tools import synth
from conftest import load_package
P = load_package()
n = int(sys.argv[1]); G = int(sys.argv[2])
w = synth.make_world(seed=2, G=G, k=1, n_mut=3)
t = time.time(); u0 = synth.make_long_batch(w, 250, seed=5, len_lo=9000, len_hi=11000); print('generated 250 reads in %.1f s' % (time.time() - t))
# replicate the 250 generated reads to n (the kernels do not care that reads repeat)
rep = (n + 249) // 250
def tile_off(o, k): return np.concatenate([[0]] + [o[1:] + i * o[-1] for i in range(k)]).astype(np.int32)
u = dict(n_pairs=250 * rep, read_off=tile_off(u0['read_off'], rep), read_bases=np.tile(u0['read_bases'], rep), read_quals=np.tile(u0['read_quals'], rep),
         chain_off=tile_off(u0['chain_off'], rep), read_primary=np.concatenate([u0['read_primary'] + i * u0['n_chains'] for i in range(rep)]).astype(np.int32),
         n_chains=u0['n_chains'] * rep, chain_contig=np.tile(u0['chain_contig'], rep), chain_pos=np.tile(u0['chain_pos'], rep), chain_offset=np.tile(u0['chain_offset'], rep),
         chain_as=np.tile(u0['chain_as'], rep), chain_reverse=np.tile(u0['chain_reverse'], rep), cigar_off=tile_off(u0['cigar_off'], rep), cigar=np.tile(u0['cigar'], rep))
ctx = P.Context(w["graph"], w["contigs"], insert_mean=200.0, insert_sd=35.0, rng_seed=3, long_read_mode=1, max_columns=16384)
gb = ctx.batch_unpaired(u)
for it in range(2):
    t = time.perf_counter(); gb.align(); st = gb.stats(); dt = time.perf_counter() - t
    print('run %d: %d reads (%.1f Mbases): project %.1f ms, pad+score %.1f ms, select %.1f ms -> %.0f reads/s, %.1f Mbases/s; errors %d' % (
        it, u['n_pairs'], u['read_off'][-1] / 1e6, st.ms_project, st.ms_extend, st.ms_pair, u['n_pairs'] / dt, u['read_off'][-1] / dt / 1e6, st.n_errors))
