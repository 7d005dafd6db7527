"""Insert-size estimation (mapper/processBAM.cpp:991-1165) on the projection / extension kernels."""
import ctypes as C

import numpy as np
import pytest

import oracle_binding as ob
from tools import synth


def hist(keys, counts):
    k = np.array(keys, np.int32); c = np.array(counts, np.float64); m = C.c_double(); s = C.c_double()
    rc = ob.lib().orc_insert_size_from_histogram(len(k), k.ctypes.data_as(C.POINTER(C.c_int32)), c.ctypes.data_as(C.POINTER(C.c_double)), C.byref(m), C.byref(s))
    assert rc == 0
    return m.value, s.value


def test_oracle_histogram_hand_derived(oracle):
    # total 10: cumulative 1, 3, 6, 9, 10 -> 20 % point (>= 2) at key 110, median (>= 5) at 120, 80 % (>= 8) at 130; sd = max(|120-110|, |120-130|) = 10
    assert hist([100, 110, 120, 130, 200], [1, 2, 3, 3, 1]) == (120.0, 10.0)
    # asymmetric: 80 % point far out -> it sets the sd; fractional weights as produced by pairs with several underlying sequences
    assert hist([90, 100, 300], [0.5, 2.0, 1.0]) == (100.0, 200.0)
    # a single key: everything coincides
    assert hist([250], [7.5]) == (250.0, 0.0)


@pytest.mark.gpu
@pytest.mark.parametrize("seed,G,k,n_pairs", [(1, 6000, 1, 500), (3, 8000, 3, 400)], ids=["seed1", "seed3"])
def test_insert_size_matches_oracle(pkg, oracle, seed, G, k, n_pairs):
    w = synth.make_world(seed=seed, G=G, k=k)
    b = synth.make_batch(w, n_pairs, seed=seed + 10)
    # the context's own insert-size parameters are placeholders here: they are what is being estimated
    o = oracle(w["graph"], w["contigs"], insert_mean=1.0, insert_sd=1.0, rng_seed=31)
    e = o.estimate_insert_size(b)
    ctx = pkg.Context(w["graph"], w["contigs"], insert_mean=1.0, insert_sd=1.0, rng_seed=31)
    g = ctx.estimate_insert_size(b)
    assert g == e
    assert e["n_used"] == n_pairs and 0 <= e["n_skipped"] < n_pairs // 4
    # the generator draws inner distances around insert_mean with sd insert_sd: the robust estimate lands close to them
    assert abs(e["mean"] - b["insert_mean"]) < 15 and 0.5 * b["insert_sd"] < e["sd"] < 2.0 * b["insert_sd"]
