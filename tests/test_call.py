"""The call of one locus (hla/HLATyper.cpp:2366-2541): sort, posteriors, marginals, first / second allele."""
import numpy as np
import pytest

import oracle_binding as ob


def tri(c1, c2, C):
    return c1 * C - c1 * (c1 - 1) // 2 + (c2 - c1)


def table(C, rng, dup=()):
    """Synthetic all-pairs table from per-cluster read likelihood rows (so that LL(c1,c2) has the structure of the real one);
    clusters listed in `dup` copy another cluster's rows: exact ties in LL and mismatches."""
    R = 40
    ll = -rng.random((C, R)) * 30 - 1; mm = rng.integers(0, 4, (C, R))
    for a, b in dup:
        ll[a] = ll[b]; mm[a] = mm[b]
    nP = C * (C + 1) // 2
    pairLL = np.zeros(nP); misAvg = np.zeros(nP); misMin = np.zeros(nP)
    for c1 in range(C):
        for c2 in range(c1, C):
            i = tri(c1, c2, C)
            hi = np.maximum(ll[c1], ll[c2]); lo = np.minimum(ll[c1], ll[c2])
            pairLL[i] = np.sum(np.log(0.5) + hi + np.log1p(np.exp(lo - hi)))
            misAvg[i] = np.sum((mm[c1] + mm[c2]) / 2.0); misMin[i] = np.sum(np.minimum(mm[c1], mm[c2]))
    return pairLL, misAvg, misMin


def table_rows(C, rng, dup=(), R=40):
    """The same table, one row of pairs (c1, c1 .. C-1) at a time: usable at the size of a real locus (C = 3000 .. 5000)."""
    ll = -rng.random((C, R)) * 30 - 1; mm = rng.integers(0, 4, (C, R))
    for a, b in dup:
        ll[a] = ll[b]; mm[a] = mm[b]
    nP = C * (C + 1) // 2
    pairLL = np.zeros(nP); misAvg = np.zeros(nP); misMin = np.zeros(nP)
    for c1 in range(C):
        i0 = tri(c1, c1, C); n = C - c1
        hi = np.maximum(ll[c1], ll[c1:]); lo = np.minimum(ll[c1], ll[c1:])
        pairLL[i0:i0 + n] = np.sum(np.log(0.5) + hi + np.log1p(np.exp(lo - hi)), axis=1)
        misAvg[i0:i0 + n] = np.sum((mm[c1] + mm[c1:]) / 2.0, axis=1); misMin[i0:i0 + n] = np.sum(np.minimum(mm[c1], mm[c1:]), axis=1)
    return pairLL, misAvg, misMin


def test_oracle_call_hand_derived(oracle):
    # C = 3; pairs in index order: (0,0) (0,1) (0,2) (1,1) (1,2) (2,2)
    ln = np.log
    LL = np.array([ln(1.0), ln(4.0), ln(2.0), ln(1.0), ln(2.0), ln(0.5)])          # weights 1, 4, 2, 1, 2, 0.5 -> sum 10.5
    MA = np.array([3.0, 1.0, 2.0, 0.0, 1.0, 5.0]); MM = np.array([3.0, 0.0, 1.0, 0.0, 2.0, 5.0])
    r = ob.call_locus(LL, MA, MM)
    # sorted: LL descending, Mism_avg ascending among equal LL: (0,1) | (1,2) [mism 1] (0,2) [mism 2] | (1,1) [0] (0,0) [3] | (2,2)
    assert r["order"].tolist() == [1, 4, 2, 3, 0, 5] and r["n_sort_ties"] == 0
    assert r["max_pair"] == 1 and np.isclose(r["ll_max"], ln(4.0))
    assert np.allclose(r["p_normalized"], np.array([1, 4, 2, 1, 2, 0.5]) / 10.5, rtol=1e-15)
    # marginals: cluster 0 = (1+4+2)/10.5, cluster 1 = (4+1+2)/10.5, cluster 2 = (2+2+0.5)/10.5 -> clusters 0 and 1 tie: the first wins
    assert np.allclose(r["cluster_marginal"], np.array([7, 7, 4.5]) / 10.5, rtol=1e-15)
    assert r["first_cluster"] == 0
    # second allele: pairs with cluster 0: (0,0) P=1, (0,1) P=4, (0,2) P=2 -> cluster 1
    assert r["second_cluster"] == 1 and np.isclose(r["second_p"], 4 / 10.5)
    # ties of the second allele are resolved by the smallest Mism_min, then by the smallest cluster
    LL2 = np.array([ln(1.0), ln(4.0), ln(4.0), ln(1.0), ln(1.0), ln(1.0)])
    assert ob.call_locus(LL2, MA, np.array([0.0, 2.0, 1.0, 0.0, 0.0, 0.0]))["second_cluster"] == 2
    assert ob.call_locus(LL2, MA, np.array([0.0, 1.0, 1.0, 0.0, 0.0, 0.0]))["second_cluster"] == 1


def same_up_to_ties(order_a, order_b, LL, MA):
    """Equal permutations except inside runs of pairs that are equal in both sort keys (unspecified in the reference)."""
    if np.array_equal(order_a, order_b):
        return True
    ka = np.stack([LL[order_a], MA[order_a]], 1); kb = np.stack([LL[order_b], MA[order_b]], 1)
    return np.array_equal(ka, kb) and np.array_equal(np.sort(order_a), np.sort(order_b))


@pytest.mark.gpu
@pytest.mark.parametrize("C,dup", [(1, ()), (2, ()), (37, ()), (200, ()), (120, ((5, 17), (40, 3), (41, 3), (119, 0))), (3000, ((5, 17), (2999, 0), (1500, 1499), (1501, 1499))), (5000, ())],
                         ids=["C1", "C2", "C37", "C200", "ties", "C3000-ties", "C5000"])
def test_call_matches_oracle(pkg, oracle, C, dup):
    from tools import synth
    rng = np.random.default_rng(100 + C)
    LL, MA, MM = table(C, rng, dup) if C <= 200 else table_rows(C, rng, dup)
    e = ob.call_locus(LL, MA, MM)
    w = synth.make_world(seed=1, G=300, k=1); ctx = pkg.Context(w["graph"], w["contigs"])
    g = ctx.call_locus(LL, MA, MM)
    for k in ("first_cluster", "second_cluster", "max_pair", "n_sort_ties"):
        assert g[k] == e[k], k
    assert same_up_to_ties(g["order"], e["order"], LL, MA)
    if dup:
        assert e["n_sort_ties"] > 0
    else:
        assert np.array_equal(g["order"], e["order"])
    # posteriors: device exp and a tree-shaped normalising sum, tolerance 1e-9 relative (north_star: 1e-6)
    assert np.allclose(g["p_normalized"], e["p_normalized"], rtol=1e-9, atol=1e-300)
    assert np.allclose(g["cluster_marginal"], e["cluster_marginal"], rtol=1e-9, atol=1e-300)
    assert np.isclose(g["first_marginal"], e["first_marginal"], rtol=1e-9) and np.isclose(g["second_p"], e["second_p"], rtol=1e-9)
    assert g["ll_max"] == e["ll_max"]
    # clusters with identical likelihoods keep bit-identical marginals (same sequence of additions): the first one wins as in the reference
    for a, b in dup:
        assert g["cluster_marginal"][a] == g["cluster_marginal"][b]
