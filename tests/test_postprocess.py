"""Per-pair post-processing (processBAM.cpp:2411-2446): coverage counters and includeInHLA."""
import numpy as np
import pytest

import oracle_binding as ob
from tools import synth


def _pairs(levels_per_mate, gchars_per_mate, stride=16, status=None):
    n = len(levels_per_mate) // 2
    d = dict(pair_status=np.zeros(n, np.int32) if status is None else np.array(status, np.int32), n_cols=np.zeros(2 * n, np.int32),
             col_level=np.zeros(2 * n * stride, np.int32), col_gchar=np.zeros(2 * n * stride, np.uint8))
    for r, (lv, g) in enumerate(zip(levels_per_mate, gchars_per_mate)):
        d["n_cols"][r] = len(lv); d["col_level"][r * stride: r * stride + len(lv)] = lv
        d["col_gchar"][r * stride: r * stride + len(lv)] = np.frombuffer(g.encode(), np.uint8)
    return d, n, stride


def test_oracle_postprocess_hand_derived(oracle):
    # pair 0: mate 1 covers levels 5,6,(gap at 7),8 with one insertion column; mate 2 covers 20..22
    # pair 1: flagged pair (status < 0): contributes nothing; pair 2: an unaligned mate (all levels -1) and a mate at 40..41
    d, n, stride = _pairs([[5, 6, -1, 7, 8], [20, 21, 22], [1, 2], [3], [-1, -1], [40, 41]],
                          ["AC_" + "_G", "TTT", "AA", "C", "__", "GG"], status=[0, -2, 0])
    #                                   ^ level 7 has graph char '_' (a deletion in the graph): not counted; the insertion column has level -1
    gf, gl = [8, 100], [19, 200]             # gene 0 = [8, 19]: touches mate 1 of pair 0 at its last level (closed interval); nothing near 40
    cov, inc = ob.postprocess_pairs(d, n, stride, gf, gl, 64)
    exp = np.zeros(64, np.int32); exp[[5, 6, 8, 20, 21, 22, 40, 41]] = 1
    assert np.array_equal(cov, exp)
    assert inc.tolist() == [1, 0, 0]
    # interval ends are inclusive on both sides (IntervalTree.h:166); a gene strictly between the mates is not hit
    assert ob.postprocess_pairs(d, n, stride, [9], [19], 64)[1].tolist() == [0, 0, 0]
    assert ob.postprocess_pairs(d, n, stride, [0], [5], 64)[1].tolist() == [1, 0, 0]
    assert ob.postprocess_pairs(d, n, stride, [41], [41], 64)[1].tolist() == [0, 0, 1]
    # counters accumulate over calls
    cov2, _ = ob.postprocess_pairs(d, n, stride, gf, gl, 64, cov=cov)
    assert np.array_equal(cov2, 2 * exp)


@pytest.mark.gpu
@pytest.mark.parametrize("seed,G,k,n_pairs", [(1, 5000, 1, 300), (2, 8000, 0, 150)], ids=["seed1", "seed2"])
def test_postprocess_matches_oracle(pkg, oracle, seed, G, k, n_pairs):
    w = synth.make_world(seed=seed, G=G, k=k)
    b = synth.make_batch(w, n_pairs, seed=seed + 10)
    o = oracle(w["graph"], w["contigs"], insert_mean=b["insert_mean"], insert_sd=b["insert_sd"], rng_seed=777)
    exp_pairs = o.align_batch(b)["pairs"]
    rng = np.random.default_rng(seed)
    gf = np.sort(rng.integers(0, G - 300, 12)).astype(np.int32); gl = (gf + rng.integers(1, 250, 12)).astype(np.int32)
    n_cov = int(w["graph"]["n_levels"]) - 1           # bases_per_level has NodesPerLevel.size() - 1 entries (processBAM.cpp:1867)
    cov_e, inc_e = ob.postprocess_pairs(exp_pairs, b["n_pairs"], o.max_columns, gf, gl, n_cov)
    ctx = pkg.Context(w["graph"], w["contigs"], insert_mean=b["insert_mean"], insert_sd=b["insert_sd"], rng_seed=777)
    ctx.set_gene_intervals(gf, gl)
    gb = ctx.batch(b); gb.align()
    inc = gb.postprocess()
    cov = ctx.coverage()
    assert np.array_equal(inc, inc_e) and 0 < inc.sum() < len(inc)
    assert np.array_equal(cov, cov_e) and cov.sum() > 0
    # the counters accumulate over batches and can be reset
    gb.postprocess()
    assert np.array_equal(ctx.coverage(reset=True), 2 * cov_e)
    assert ctx.coverage().sum() == 0
    # no gene intervals: nothing is included
    ctx.set_gene_intervals([], [])
    assert gb.postprocess().sum() == 0
