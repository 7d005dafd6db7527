"""HLATyper per-read scoring: oracle by hand on a tiny locus (CPU) and GPU kernels vs oracle."""
import math

import numpy as np
import pytest

import oracle_binding as ob
from tools import synth


def _tiny():
    # 2 clusters over 4 exon columns; one read with: match, mismatch, deletion against a base, insertion (len 2) against '_'
    seq = np.frombuffer(b"ACG_" + b"ATG_", np.uint8)
    return dict(n_clusters=2, exon_length=4, cluster_seq=seq, n_reads=1, pos_off=np.array([0, 4], np.int32),
                pos_exon=np.array([0, 1, 2, 3], np.int32), pos_g0=np.frombuffer(b"AC_T", np.uint8), pos_glen=np.array([1, 1, 1, 2], np.int32),
                pos_qual=np.array([33 + 30] * 4, np.uint8), pos_use=np.array([1, 1, 1, 1], np.uint8))


def test_exon_loglik_by_hand(oracle):
    LL, mism = ob.exon_loglik(_tiny())
    r = 0.001; p = 1 - 10 ** (-3.0); p = min(p, 0.999)
    lmm, ldel, lins = math.log(1 - 2 * r), math.log(r), math.log(r) + math.log(0.25)
    # cluster 0 "ACG_": A=A match, C=C match, '_' vs G deletion, "TN" vs '_' insertion of 2
    exp0 = (lmm + math.log(p)) + (lmm + math.log(p)) + ldel + 2 * lins
    # cluster 1 "ATG_": second column C vs T mismatch
    exp1 = (lmm + math.log(p)) + (lmm + math.log((1 - p) / 3)) + ldel + 2 * lins
    assert LL[0, 0] == pytest.approx(exp0, rel=1e-14) and LL[1, 0] == pytest.approx(exp1, rel=1e-14)
    # mismatches: genotype != "_" and != exon char: cluster 0: insertion column only; cluster 1: + the C/T column
    assert mism[:, 0].tolist() == [1, 2]


def test_pair_loglik_by_hand(oracle):
    LL = np.array([[-1.0, -2.0], [-3.0, -2.0]]); mism = np.array([[0, 2], [1, 2]], np.int32)
    pl, ma, mn = ob.pair_loglik(LL, mism)
    la = lambda a, b: math.log(0.5) + math.log(1 + math.exp(-abs(a - b))) + max(a, b)
    assert pl[0] == pytest.approx(la(-1, -1) + la(-2, -2), rel=1e-14)        # (0,0): logAvg(a,a) = a
    assert pl[1] == pytest.approx(la(-1, -3) + la(-2, -2), rel=1e-14)        # (0,1)
    assert pl[2] == pytest.approx(-3 + -2, rel=1e-14)                         # (1,1)
    assert ma.tolist() == [2.0, 2.5, 3.0] and mn.tolist() == [2.0, 2.0, 3.0]


@pytest.mark.gpu
# (3000, 400) and (5000, 1000): the size of a real class-I / class-II locus of PRG_MHC_GRCh38_withIMGT -- 12 and 20 column tiles of the tiled all-pairs kernel,
# 4.5 M and 12.5 M cluster pairs (the oracle computes its rows on all host cores; each sum keeps the reference's order, hla/HLATyper.cpp:2293-2364)
@pytest.mark.parametrize("C,R,seed", [(200, 300, 5), (37, 1500, 6), (1, 5, 7), (513, 64, 8), (3000, 400, 9), (5000, 1000, 10)])
def test_typer_kernels_match_oracle(pkg, oracle, C, R, seed):
    loc = synth.make_locus(seed=seed, n_clusters=C, n_reads=R)
    w = synth.make_world(seed=1, G=300, k=1)
    ctx = pkg.Context(w["graph"], w["contigs"])
    eLL, em = ob.exon_loglik(loc)
    gLL, gm = ctx.exon_loglik(loc)
    assert np.array_equal(gm, em)
    assert np.allclose(gLL, eLL, rtol=1e-12, atol=0) and np.array_equal(gLL, eLL)      # table-driven, reference order: bit-identical
    epl, ema, emn = ob.pair_loglik(eLL, em)
    gpl, gma, gmn = ctx.pair_loglik(gLL, gm)
    assert np.array_equal(gma, ema) and np.array_equal(gmn, emn)
    assert np.allclose(gpl, epl, rtol=1e-9, atol=0)                                   # north star: summed log-likelihoods within 1e-6 relative
    # the best pair is the same
    assert int(np.argmax(gpl)) == int(np.argmax(epl))


@pytest.mark.gpu
@pytest.mark.parametrize("C,R,seed", [(200, 300, 5), (1, 5, 7), (513, 64, 8), (3000, 400, 9)])
def test_type_locus_equals_the_three_calls(pkg, C, R, seed):
    """hlala_type_locus (the per-read and all-pairs tables stay on the device between the steps) returns, bit for bit, what hlala_exon_loglik ->
    hlala_pair_loglik -> hlala_call_locus return one after the other (hla/HLATyper.cpp:2067-2541)."""
    loc = synth.make_locus(seed=seed, n_clusters=C, n_reads=R)
    w = synth.make_world(seed=1, G=300, k=1)
    ctx = pkg.Context(w["graph"], w["contigs"])
    LL, mism = ctx.exon_loglik(loc)
    pl, ma, mn = ctx.pair_loglik(LL, mism)
    call = ctx.call_locus(pl, ma, mn)
    for want_table in (True, False):
        got = ctx.type_locus(loc, want_reads_table=want_table)
        if want_table:
            assert np.array_equal(got["LL"], LL) and np.array_equal(got["mism"], mism)
        else:
            assert got["LL"] is None
        assert np.array_equal(got["pairLL"], pl) and np.array_equal(got["misAvg"], ma) and np.array_equal(got["misMin"], mn)
        for k in ("order", "p_normalized", "cluster_marginal"):
            assert np.array_equal(got[k], call[k]), k
        for k in ("first_cluster", "second_cluster", "first_marginal", "second_p", "ll_max", "max_pair", "n_sort_ties"):
            assert got[k] == call[k], k


@pytest.mark.gpu
def test_type_locus_without_reads(pkg):
    loc = synth.make_locus(seed=3, n_clusters=40, n_reads=0)
    w = synth.make_world(seed=1, G=300, k=1)
    ctx = pkg.Context(w["graph"], w["contigs"])
    LL, mism = ctx.exon_loglik(loc)
    pl, ma, mn = ctx.pair_loglik(LL, mism)
    call = ctx.call_locus(pl, ma, mn)
    got = ctx.type_locus(loc)
    assert np.array_equal(got["pairLL"], pl) and np.array_equal(got["order"], call["order"]) and got["first_cluster"] == call["first_cluster"] and got["second_cluster"] == call["second_cluster"]
