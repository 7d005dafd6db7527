"""One-off stress: larger random batches than the test suite uses, product vs oracle (pairs + extended chains + work counters)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from tools import synth
from conftest import load_package
from oracle_binding import Oracle, OracleError
from util import compare_chains
from test_gpu_align import assert_pairs_equal
P = load_package()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 6000
only = set(int(x) for x in sys.argv[2].split(',')) if len(sys.argv) > 2 else None
for seed, G, k, kw, bk in ((101, 40000, 1, dict(n_mut=3), dict()), (102, 20000, 0, dict(n_largegap=2), dict(p_secondary=0.8, max_secondary=5)),
                           (103, 30000, 2, dict(extra_identical=2), dict(p_secondary=1.0, max_secondary=6, clip_max=60)), (104, 15000, 5, dict(n_mut=5), dict(indel_read_frac=0.3)),
                           (105, 25000, 1, dict(n_mut=8, mut_density=0.04), dict(p_random_secondary=0.4)),
                           (106, 20000, 1, dict(n_mut=4), dict(read_len=100, ins_mean=150.0, ins_sd=25.0, clip_max=15, indel_read_frac=0.0)),
                           (107, 30000, 2, dict(n_mut=4, n_largegap=2), dict(read_len=250, ins_mean=350.0, ins_sd=60.0, clip_max=80, indel_read_frac=0.2)),
                           (108, 15000, 1, dict(n_mut=3, gap_frac=0.6, mut_density=0.05), dict(clip_max=65, p_no_clip=0.0, p_secondary=0.9))):
    if only and seed not in only:
        continue
    t0 = time.time()
    w = synth.make_world(seed=seed, G=G, k=k, **kw)
    b = synth.make_batch(w, n, seed=seed + 1000, **bk)
    kwc = dict(insert_mean=b["insert_mean"], insert_sd=b["insert_sd"], rng_seed=seed, max_columns=384)
    try:
        exp = Oracle(w["graph"], w["contigs"], **kwc).align_batch(b)
    except OracleError as err:      # the generator made a record the reference asserts on (the product flags such a chain, the oracle stops): not comparable
        print("seed %d skipped: %s" % (seed, err), flush=True); continue
    ctx = P.Context(w["graph"], w["contigs"], **kwc)
    gb = ctx.batch(b); gb.align()
    compare_chains(gb.chains(1), exp["ext"], b["n_chains"], label="stress %d" % seed)
    assert_pairs_equal(gb.pairs(), exp["pairs"])
    st = gb.stats()
    assert (st.n_dp_calls, st.n_dp_iterations, st.n_dp_cells) == tuple(int(x) for x in exp["stats"][:3])
    print("seed %d k=%d: %d pairs, %d chains, %d DP calls (%d shared, %d re-run wider, %d large), errors %d, %.0f s" % (seed, k, n, b["n_chains"], st.n_dp_calls, st.n_dp_shared, st.n_chains_retried, st.n_dp_retried_large, st.n_errors, time.time() - t0), flush=True)
print("STRESS OK")
