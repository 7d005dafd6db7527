"""One-off stress of the smaller entry points at the reference's own sizes: insert-size estimation on 4000 pairs (extractSeeds(4000)),
the call of a locus with thousands of clusters and many exact ties, likelihood kernels at C = 3000."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from tools import synth
from conftest import load_package
import oracle_binding as ob
from oracle_binding import Oracle
from test_call import same_up_to_ties
P = load_package()
for seed, G, k in ((401, 30000, 1), (402, 20000, 0), (403, 25000, 4)):
    w = synth.make_world(seed=seed, G=G, k=k); b = synth.make_batch(w, 4000, seed=seed + 1, ins_mean=310.0, ins_sd=55.0)
    e = Oracle(w["graph"], w["contigs"], insert_mean=1.0, insert_sd=1.0, rng_seed=31).estimate_insert_size(b)
    ctx = P.Context(w["graph"], w["contigs"], insert_mean=1.0, insert_sd=1.0, rng_seed=31)
    g = ctx.estimate_insert_size(b)
    assert g == e, (g, e)
    print("insert size seed %d: mean %.3f sd %.3f used %d skipped %d" % (seed, e["mean"], e["sd"], e["n_used"], e["n_skipped"]), flush=True)
rng = np.random.default_rng(7)
for C, R, ndup in ((1500, 300, 200), (3000, 150, 600)):
    # per-cluster per-read log-likelihoods with many identical clusters (alleles that do not differ over the covered columns)
    LL = -rng.random((C, R)) * 3; M = rng.integers(0, 3, (C, R)).astype(np.int32)
    src = rng.integers(0, C, ndup); dst = rng.integers(0, C, ndup); LL[dst] = LL[src]; M[dst] = M[src]
    t0 = time.time(); pe = ob.pair_loglik(LL, M); t1 = time.time()
    pg = ctx.pair_loglik(LL, M)
    assert np.allclose(pg[0], pe[0], rtol=1e-9) and np.array_equal(pg[1], pe[1]) and np.array_equal(pg[2], pe[2])
    ce = ob.call_locus(*pe); cg = ctx.call_locus(*pe)
    for key in ("first_cluster", "second_cluster", "max_pair", "n_sort_ties"):
        assert cg[key] == ce[key], key
    assert same_up_to_ties(cg["order"], ce["order"], pe[0], pe[1])
    assert np.allclose(cg["cluster_marginal"], ce["cluster_marginal"], rtol=1e-9, atol=1e-300) and np.allclose(cg["p_normalized"], ce["p_normalized"], rtol=1e-9, atol=1e-300)
    # the call from the device's own table as well (the decisions must not depend on the last bits of the pair likelihoods)
    cg2 = ctx.call_locus(*pg)
    print("call C=%d R=%d: first %d second %d ties %d; from the device table: first %d second %d (oracle table %.0f s)" % (C, R, ce["first_cluster"], ce["second_cluster"], ce["n_sort_ties"], cg2["first_cluster"], cg2["second_cluster"], t1 - t0), flush=True)
print("MISC STRESS OK")
