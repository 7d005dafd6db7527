"""GPU parity: stage B (extendSeedChain + scoreOneAlignment) against the CPU oracle, through the C ABI."""
import numpy as np
import pytest

from tools import synth
from util import compare_chains, seeds_from_chains

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed,G,k,n_pairs", [(1, 5000, 1, 300), (2, 8000, 0, 150), (3, 8000, 3, 300), (4, 3000, 10, 200)],
                         ids=["seed1", "seed2", "seed3", "seed4"])
def test_extend_matches_oracle(pkg, oracle, seed, G, k, n_pairs):
    w = synth.make_world(seed=seed, G=G, k=k)
    b = synth.make_batch(w, n_pairs, seed=seed + 10)
    o = oracle(w["graph"], w["contigs"], insert_mean=b["insert_mean"], insert_sd=b["insert_sd"], rng_seed=777)
    seeds = seeds_from_chains(b, o.align_batch(b, stop_after_projection=True)["seeds"])
    exp = o.extend_seeds(seeds)
    ctx = pkg.Context(w["graph"], w["contigs"], insert_mean=b["insert_mean"], insert_sd=b["insert_sd"], rng_seed=777)
    gb = ctx.batch_from_seeds(seeds)
    gb.extend()
    got = gb.chains(1)
    st = gb.stats()
    assert st.n_errors == 0
    compare_chains(got, exp, seeds["n_chains"], label=f"extend seed={seed} k={k}")
    # work counters agree with the oracle's (same DP calls, iterations and evaluated cells)
    assert st.n_dp_calls == exp["_stats"][0]
    assert st.n_dp_iterations == exp["_stats"][1]
    assert st.n_dp_cells == exp["_stats"][2]
    # the gap-heavy graph (k = 0: long runs of parallel gap paths) must have gone through the wider capacity classes,
    # i.e. the parity above covers DP calls re-run by the 32-lane, 64-lane and wide kernels as well (the broad and large classes: tests/test_graph_m.py)
    print(f"k={k}: DP calls {st.n_dp_calls}, re-run in a wider class {st.n_chains_retried}, in the large class {st.n_dp_retried_large}")
    if k == 0:
        assert st.n_chains_retried > 0 and all(int(x) > 0 for x in list(st.n_dp_class)[:4]), list(st.n_dp_class)
    print(f"k={k}: DP calls sharing the DP of another chain {st.n_dp_shared}")
