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
