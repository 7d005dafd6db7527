"""One-off stress of the per-locus chain: bigger batches, several loci and graph styles, product vs oracle (exon positions, filters, likelihoods, call)."""
import os, sys, time, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from tools import synth
from conftest import load_package
import oracle_binding as ob
from oracle_binding import Oracle
P = load_package(); lib = C.CDLL(P.LIB_PATH)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
for seed, G, k, kw in ((201, 9000, 1, dict(n_mut=6, mut_density=0.03)), (202, 7000, 0, dict(n_largegap=2)), (203, 8000, 3, dict(n_mut=4))):
    t0 = time.time()
    w = synth.make_world(seed=seed, G=G, k=k, **kw)
    b = synth.make_batch(w, n, seed=seed + 1, p_secondary=0.7, max_secondary=4)
    kwc = dict(insert_mean=b["insert_mean"], insert_sd=b["insert_sd"], rng_seed=seed)
    o = Oracle(w["graph"], w["contigs"], **kwc); pe = o.align_batch(b)["pairs"]
    ctx = P.Context(w["graph"], w["contigs"], **kwc); gb = ctx.batch(b); gb.align()
    H = w["H"]; rng = np.random.default_rng(seed)
    for li in range(3):
        a0 = int(rng.integers(500, G - 2500)); e1 = (a0, a0 + 270); e2 = (a0 + 700, a0 + 976)
        lmin = e1[0]; l2e = np.full(e2[1] - e1[0], -1, np.int32); l2e[:270] = np.arange(270); l2e[700:976] = np.arange(270, 546)
        gene = (np.array([lmin], np.int32), np.array([lmin + len(l2e) - 1], np.int32))
        ctx.set_gene_intervals(*gene)
        inc_g = gb.postprocess(); _, inc_e = ob.postprocess_pairs(pe, n, o.max_columns, gene[0], gene[1], int(w["graph"]["n_levels"]) - 1)
        assert np.array_equal(inc_g, inc_e)
        eg = gb.exon_positions(lmin, l2e, b["insert_mean"], b["insert_sd"], pair_mask=inc_g)
        ee = ob.exon_positions(pe, b, o.max_columns, lmin, l2e, b["insert_mean"], b["insert_sd"], pair_mask=inc_e)
        for key in ee:
            if key == "read_reverse":
                continue
            if key == "read_mapq":
                assert np.allclose(eg[key], ee[key], rtol=1e-9, atol=1e-15); continue
            assert np.array_equal(np.asarray(eg[key]), np.asarray(ee[key])), key
        prm = P.default_filter_params(first20_n=12, high_coverage_filter=1, high_coverage_min_coverage=15, high_coverage_min_freq=0.1)
        ug, ig, sg = P.filter_positions(lib, eg, prm); ue, ie, se = ob.filter_positions(ee, prm)
        assert np.array_equal(ug, ue) and np.array_equal(ig, ie) and sg == se
        seqs = np.concatenate([H[:, e1[0]:e1[1]], H[:, e2[0]:e2[1]]], 1)
        seqs = np.unique(seqs, axis=0); Cn = len(seqs)
        xg = P.exon_in_from_positions(eg, ug, seqs, Cn, 546); xe = P.exon_in_from_positions(ee, ue, seqs, Cn, 546)
        LLg, Mg = ctx.exon_loglik(xg); LLe, Me = ob.exon_loglik(xe)
        assert np.array_equal(Mg, Me) and np.array_equal(LLg, LLe)
        pg = ctx.pair_loglik(LLg, Mg); pe2 = ob.pair_loglik(LLe, Me)
        assert np.allclose(pg[0], pe2[0], rtol=1e-9) and np.array_equal(pg[1], pe2[1]) and np.array_equal(pg[2], pe2[2])
        cg = ctx.call_locus(*pg); ce = ob.call_locus(*pe2)
        assert (cg["first_cluster"], cg["second_cluster"]) == (ce["first_cluster"], ce["second_cluster"]), (cg["first_cluster"], cg["second_cluster"], ce["first_cluster"], ce["second_cluster"])
        print("seed %d locus %d: %d reads with positions, %d positions, %d clusters, call (%d, %d), filters removed %d alleles" % (seed, li, ee["n_reads"], ee["n_pos"], Cn, ce["first_cluster"], ce["second_cluster"], se["removed_alleles"]), flush=True)
    print("seed %d done in %.0f s" % (seed, time.time() - t0), flush=True)
print("TYPER STRESS OK")
