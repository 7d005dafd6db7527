"""One-off stress of the long-read / unpaired path: more and longer reads than the test suite, product vs oracle."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from tools import synth
from conftest import load_package
from oracle_binding import Oracle
from util import compare_chains
import oracle_binding as ob
import ctypes as C
P = load_package()
PAIR_INT = ("pair_status", "best_chain", "n_combinations", "n_cols", "col_level", "col_edge", "col_gchar", "col_schar", "col_fromseed", "col_mapq")
for seed, G, k, nreads, lo, hi, ps in ((301, 60000, 1, 300, 2000, 12000, 0.0), (302, 40000, 3, 300, 600, 6000, 0.0), (303, 30000, 0, 200, 1000, 5000, 0.0), (304, 8000, 1, 600, 150, 380, 0.5)):
    t0 = time.time()
    w = synth.make_world(seed=seed, G=G, k=k)
    u = synth.make_long_batch(w, nreads, seed=seed + 1, len_lo=lo, len_hi=hi, p_second=ps)
    cols = 16384 if hi > 500 else 1024        # (max_columns bounds the intermediate columns of the projection too: keep headroom)
    kw = dict(insert_mean=200.0, insert_sd=35.0, rng_seed=3, long_read_mode=1, max_columns=cols)
    e = Oracle(w["graph"], w["contigs"], **kw).align_long_reads(u)
    ctx = P.Context(w["graph"], w["contigs"], **kw)
    gb = ctx.batch_unpaired(u); gb.align()
    compare_chains(gb.chains(1), e["ext"], u["n_chains"], check_dp=False, label="long %d" % seed)
    g = gb.pairs(); x = e["pairs"]; n = nreads
    # documented capacity: a read with several alignments of which one has more than 512 columns is flagged (include/hlala_gpu.h);
    # long-read BAMs carry primaries only (processBAM.cpp:725-738), so such reads exist in this synthetic set only
    nch = np.diff(u["chain_off"]); mx = np.array([e["ext"]["n_cols"][u["chain_off"][r]:u["chain_off"][r + 1]].max() for r in range(n)])
    lim = (nch > 1) & (mx > 512)
    assert (np.asarray(g["pair_status"])[:n][lim] == -1).all()
    keep = ~lim
    for key in PAIR_INT:
        a = np.asarray(g[key]); b2 = np.asarray(x[key])
        if key in ("pair_status", "best_chain", "n_combinations", "n_cols"):
            assert np.array_equal(a[:n][keep], b2[:n][keep]), key
        else:
            assert np.array_equal(a[:n * cols].reshape(n, cols)[keep], b2[:n * cols].reshape(n, cols)[keep]), key
    assert np.allclose(g["pair_ll"][:n][keep], x["pair_ll"][:n][keep], rtol=1e-12, atol=0) and np.allclose(g["mate_mapq"][:n][keep], x["mate_mapq"][:n][keep], rtol=1e-9)
    # per-read exon positions and filters for three loci (unpaired path of the typer), only where nothing was flagged
    npos = 0
    if not lim.any():
        rng = np.random.default_rng(seed); lib = C.CDLL(P.LIB_PATH)
        for li in range(3):
            a0 = int(rng.integers(200, G - 1500)); l2e = np.full(1000, -1, np.int32); l2e[:300] = np.arange(300); l2e[650:950] = np.arange(300, 600)
            eg = gb.exon_positions(a0, l2e, 0, 0, min_alignment_columns=min(1000, lo))
            ee = ob.exon_positions(x, u, cols, a0, l2e, 0, 0, unpaired=True, min_alignment_columns=min(1000, lo))
            for key in ee:
                if key == "read_reverse":
                    continue
                if key == "read_mapq":
                    assert np.allclose(eg[key], ee[key], rtol=1e-9, atol=1e-15); continue
                assert np.array_equal(np.asarray(eg[key]), np.asarray(ee[key]), equal_nan=True) if isinstance(ee[key], np.ndarray) and ee[key].dtype.kind == "f" else np.array_equal(eg[key], ee[key]), key
            prm = P.default_filter_params(first20_n=8)
            ug, ig, sg = P.filter_positions(lib, eg, prm); ue, ie, se = ob.filter_positions(ee, prm)
            assert np.array_equal(ug, ue) and np.array_equal(ig, ie) and sg == se
            npos += ee["n_pos"]
    print("   exon positions compared:", npos)
    print("seed %d k=%d: %d reads, max columns %d, errors %d, %.0f s" % (seed, k, n, int(np.asarray(x["n_cols"])[:n].max()), gb.stats().n_errors, time.time() - t0), flush=True)
print("LONG STRESS OK")
