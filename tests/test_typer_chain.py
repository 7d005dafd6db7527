"""GPU: the whole per-locus chain after alignment, against the same chain on the oracle --
postprocess (includeInHLA) -> exon positions -> filters -> per-cluster x per-read likelihoods -> all pairs -> call."""
import ctypes as C

import numpy as np
import pytest

import oracle_binding as ob
from tools import synth

pytestmark = pytest.mark.gpu


def test_locus_chain_matches_oracle(pkg, oracle):
    G = 6000
    w = synth.make_world(seed=11, G=G, k=1)
    b = synth.make_batch(w, 700, seed=12)
    o = oracle(w["graph"], w["contigs"], insert_mean=b["insert_mean"], insert_sd=b["insert_sd"], rng_seed=5)
    pe = o.align_batch(b)["pairs"]
    ctx = pkg.Context(w["graph"], w["contigs"], insert_mean=b["insert_mean"], insert_sd=b["insert_sd"], rng_seed=5)
    gb = ctx.batch(b); gb.align()
    # one "gene" with two exons in the middle of the graph
    lmin = 2000; l2e = np.full(1300, -1, np.int32); l2e[100:400] = np.arange(300); l2e[800:1100] = np.arange(300, 600); P_ex = 600
    n_cov = int(w["graph"]["n_levels"]) - 1
    gene = (np.array([lmin], np.int32), np.array([lmin + len(l2e) - 1], np.int32))
    ctx.set_gene_intervals(*gene)
    inc_g = gb.postprocess()
    _, inc_e = ob.postprocess_pairs(pe, b["n_pairs"], o.max_columns, gene[0], gene[1], n_cov)
    assert np.array_equal(inc_g, inc_e) and 20 < inc_e.sum() < b["n_pairs"]
    eg = gb.exon_positions(lmin, l2e, b["insert_mean"], b["insert_sd"], pair_mask=inc_g)
    ee = ob.exon_positions(pe, b, o.max_columns, lmin, l2e, b["insert_mean"], b["insert_sd"], pair_mask=inc_e)
    for k in ee:
        if k == "read_reverse":                     # the oracle entry point is not handed the strands: checked against the batch below
            continue
        if k == "read_mapq":                        # posteriors (device exp()): the tolerance of test_gpu_align
            assert np.allclose(eg[k], ee[k], rtol=1e-9, atol=1e-15); continue
        assert np.array_equal(np.asarray(eg[k]), np.asarray(ee[k]), equal_nan=True) if isinstance(ee[k], np.ndarray) and ee[k].dtype.kind == "f" else np.array_equal(eg[k], ee[k]), k
    assert ee["n_reads"] > 30
    assert np.array_equal(eg["read_reverse"], b["chain_reverse"][pe["best_chain"].reshape(-1, 2)[eg["read_pair"]].reshape(-1)])
    # low coverage here: let the first-N filter act on positions with >= 6 reads so that it does something
    prm = pkg.default_filter_params(first20_n=6, first20_limit_per_read=0)
    ug, ig, sg = pkg.filter_positions(C.CDLL(pkg.LIB_PATH), eg, prm)
    ue, ie, se = ob.filter_positions(ee, prm)
    assert np.array_equal(ug, ue) and np.array_equal(ig, ie) and sg == se and se["considered_positions"] > 0
    # clusters: random exon strings around a consensus
    rng = np.random.default_rng(3); Cn = 60
    nuc = np.frombuffer(b"ACGT", np.uint8); cons = nuc[rng.integers(0, 4, P_ex)]
    seqs = np.tile(cons, (Cn, 1)); snp = rng.random((Cn, P_ex)) < 0.03; seqs[snp] = nuc[rng.integers(0, 4, int(snp.sum()))]
    xin_g = pkg.exon_in_from_positions(eg, ug, seqs, Cn, P_ex); xin_e = pkg.exon_in_from_positions(ee, ue, seqs, Cn, P_ex)
    LLg, Mg = ctx.exon_loglik(xin_g); LLe, Me = ob.exon_loglik(xin_e)
    assert np.array_equal(Mg, Me) and np.array_equal(LLg, LLe)                        # table-driven: bit-identical
    pg = ctx.pair_loglik(LLg, Mg); pe2 = ob.pair_loglik(LLe, Me)
    assert np.allclose(pg[0], pe2[0], rtol=1e-9) and np.array_equal(pg[1], pe2[1]) and np.array_equal(pg[2], pe2[2])
    cg = ctx.call_locus(*pe2); ce = ob.call_locus(*pe2)                                # same table in: the call itself
    assert (cg["first_cluster"], cg["second_cluster"]) == (ce["first_cluster"], ce["second_cluster"])
    assert np.allclose(cg["cluster_marginal"], ce["cluster_marginal"], rtol=1e-9, atol=1e-300)
