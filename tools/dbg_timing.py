import sys, time, os, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from tools import synth
from conftest import load_package
P = load_package()
n_pairs = int(sys.argv[1]); G = int(sys.argv[2])
# usage: dbg_timing.py <pairs> <levels> [m [frac_gene]]   (m: Graph M instead of the round-1 stand-in)
if len(sys.argv) > 3 and sys.argv[3] == "m":
    w = synth.make_world_m(seed=2, n_levels=G)
    b = synth.make_batch_m(w, n_pairs, seed=1000, frac_gene=float(sys.argv[4]) if len(sys.argv) > 4 else 0.3)
else:
    w = synth.make_world(seed=2, G=G, k=1, n_mut=3)
    b = synth.make_batch_fast(w, n_pairs, seed=1000)
ctx = P.Context(w["graph"], w["contigs"], insert_mean=b["insert_mean"], insert_sd=b["insert_sd"], rng_seed=12345)
gb = ctx.batch(b)
gb.align(); gb.stats()
gb.align(); st = gb.stats()
buf = (C.c_ulonglong * 32)()
ctx.lib.hlala_debug_counters.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_ulonglong)]
ctx.lib.hlala_debug_counters(ctx.h, gb.b, buf)
d = np.array(list(buf)[8:16], dtype=np.float64)
if d[6] > 0:
    print('k_dp<tiny> cycles/trip: fetch %.0f done %.0f expand %.0f bt %.0f select %.0f run %.0f | trips %d, run groups/trip %.2f' % (d[0]/d[6], d[1]/d[6], d[2]/d[6], d[3]/d[6], d[4]/d[6], d[5]/d[6], d[6], d[7]/d[6]))
print('retry ms', st.ms_extend_retry, 'dp main ms', st.ms_dp_main, 'retried', st.n_chains_retried, 'retried large', st.n_dp_retried_large, 'errors', st.n_errors); print('ms', st.ms_project, st.ms_extend, st.ms_pair, 'pairs/s', n_pairs/((st.ms_project+st.ms_extend+st.ms_pair)*1e-3))

pj = np.array(list(buf)[16:24], dtype=np.float64)
if pj[7] > 0:
    # counters[16..23]: the projection's phase clocks (HLALA_DEBUG=1, default build) or, in the -DHLALA_DP_TIMING build, k_stitch_chains' (that build leaves the projection's out)
    if False:
        print('stitch cycles/chain (timing build): fetch+status %.0f descriptors %.0f stitch %.0f LL %.0f firstlast+out %.0f | chains %d' % tuple(list(pj[:5] / pj[7]) + [int(pj[7])]))
    else:
        print('project cycles/chain: walk %.0f trim+pad %.0f clean %.0f restrict %.0f stage+dp %.0f backtrace %.0f | chains %d' % tuple(list(pj[:6] / pj[7]) + [int(pj[7])]))
hh = np.array(list(buf)[24:32], dtype=np.float64)
if pj[7] > 0 and d[6] == 0 and hh[2] > 0:
    print('project, chunked form: chunk staging %.0f and level loops %.0f cycles/chain (all chains); %.2f chunks/chain, %.1f levels/chunk, %.0f cycles/level' % (hh[0] / pj[7], hh[1] / pj[7], hh[2] / pj[7], hh[3] / hh[2], hh[1] / hh[3]))
    hh[:] = 0
if hh[0] > 0 and d[6] > 0 and pj[0] > 0:
    # -DHLALA_DP_TIMING, round 4: the evaluate pass in pieces (counters[16..21] = tPh[8..13])
    print('evaluate pieces, cycles/trip: reads+decode %.0f scan+slot %.0f back pointers %.0f new cell / early / complete %.0f existing %.0f diff+stash %.0f (rest of the pass: loop ends, fences)' % tuple(pj[:6] / d[6]))
if hh[0] > 0 and d[6] > 0:
    print('dp_iterate cycles/trip (group 0 of each wave; waitcnt(0) before each clock): header+records %.0f pushes %.0f tlist %.0f early-lookup %.0f evaluate-passes %.0f post-evaluate %.0f filter+writeback %.0f' % (hh[4]/d[6], hh[5]/d[6], hh[0]/d[6], hh[3]/d[6], hh[6]/d[6], hh[1]/d[6], hh[2]/d[6]))
