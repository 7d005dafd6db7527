"""Long-read / unpaired mode (SURVEY a14 / a15 unpaired variants; mapper/processBAM.cpp:3618-3838, 3900-4059)."""
import numpy as np
import pytest

from tools import synth
from util import compare_chains

pytestmark = pytest.mark.gpu

PAIR_INT = ("pair_status", "best_chain", "n_combinations", "n_cols", "col_level", "col_edge", "col_gchar", "col_schar", "col_fromseed", "col_mapq")


@pytest.mark.parametrize("seed,G,k,n_pairs,long_mode", [(1, 5000, 1, 200, 1), (3, 8000, 3, 150, 0), (2, 8000, 0, 100, 1)], ids=["seed1-long", "seed3-short", "seed2-long"])
def test_unpaired_matches_oracle(pkg, oracle, seed, G, k, n_pairs, long_mode):
    w = synth.make_world(seed=seed, G=G, k=k)
    u = synth.as_unpaired(synth.make_batch(w, n_pairs, seed=seed + 10))
    n = u["n_pairs"]
    o = oracle(w["graph"], w["contigs"], insert_mean=200.0, insert_sd=35.0, rng_seed=777, long_read_mode=long_mode)
    e = o.align_long_reads(u)
    ctx = pkg.Context(w["graph"], w["contigs"], insert_mean=200.0, insert_sd=35.0, rng_seed=777, long_read_mode=long_mode)
    gb = ctx.batch_unpaired(u); gb.align()
    st = gb.stats()
    assert st.n_errors == 0 and st.n_dp_calls == 0                       # alignOneLongRead never runs the extension DP
    compare_chains(gb.chains(0), e["seeds"], u["n_chains"], check_ll=False, check_dp=False, label="unpaired seeds")
    compare_chains(gb.chains(1), e["ext"], u["n_chains"], check_dp=False, label="unpaired padded chains")
    g = gb.pairs(); x = e["pairs"]; stride = o.max_columns
    for key in PAIR_INT:
        per = {"pair_status": n, "best_chain": n, "n_combinations": n, "n_cols": n}.get(key, n * stride)
        assert np.array_equal(np.asarray(g[key])[:per], np.asarray(x[key])[:per]), key
    assert np.allclose(g["pair_ll"][:n], x["pair_ll"][:n], rtol=1e-12, atol=0)
    assert np.allclose(g["pair_mapq"][:n], x["pair_mapq"][:n], rtol=1e-9) and np.allclose(g["mate_mapq"][:n], x["mate_mapq"][:n], rtol=1e-9)
    assert (np.asarray(x["n_combinations"])[:n] > 1).any()                # some reads have several chains: the posterior path is exercised


def test_unpaired_exon_positions_and_filters(pkg, oracle):
    import ctypes as C
    import oracle_binding as ob
    G = 6000
    w = synth.make_world(seed=5, G=G, k=1)
    u = synth.as_unpaired(synth.make_batch(w, 300, seed=6))
    n = u["n_pairs"]
    o = oracle(w["graph"], w["contigs"], insert_mean=200.0, insert_sd=35.0, rng_seed=9, long_read_mode=1)
    x = o.align_long_reads(u)["pairs"]
    ctx = pkg.Context(w["graph"], w["contigs"], insert_mean=200.0, insert_sd=35.0, rng_seed=9, long_read_mode=1)
    gb = ctx.batch_unpaired(u); gb.align()
    lmin = 1500; l2e = np.full(1600, -1, np.int32); l2e[50:700] = np.arange(650); l2e[900:1500] = np.arange(650, 1250)
    # coverage counters / includeInHLA per read
    ctx.set_gene_intervals([lmin], [lmin + len(l2e) - 1])
    inc = gb.postprocess()
    xs = {k: v for k, v in x.items()}
    # the oracle routine is written for pairs: present every read as the first mate of a pair whose second mate has no columns
    stride = o.max_columns
    two = dict(pair_status=xs["pair_status"][:n], n_cols=np.stack([xs["n_cols"][:n], np.zeros(n, np.int32)], 1).reshape(-1),
               col_level=np.concatenate([xs["col_level"][:n * stride].reshape(n, stride), np.zeros((n, stride), np.int32)], 1).reshape(-1),
               col_gchar=np.concatenate([xs["col_gchar"][:n * stride].reshape(n, stride), np.zeros((n, stride), np.uint8)], 1).reshape(-1))
    cov_e, inc_e = ob.postprocess_pairs(two, n, stride, [lmin], [lmin + len(l2e) - 1], int(w["graph"]["n_levels"]) - 1)
    assert np.array_equal(inc, inc_e) and np.array_equal(ctx.coverage(), cov_e) and inc.sum() > 10
    # exon positions: reads are 150 columns long here, so the length threshold of the reference (1000) is lowered to let them through
    for mac in (100, 1000):
        g = gb.exon_positions(lmin, l2e, 0, 0, min_mapq=0.3, pair_mask=inc, min_alignment_columns=mac)
        e = ob.exon_positions(x, u, stride, lmin, l2e, 0, 0, min_mapq=0.3, pair_mask=inc_e, unpaired=True, min_alignment_columns=mac)
        for key in e:
            if key == "read_reverse":
                assert np.array_equal(g[key][0::2], u["chain_reverse"][np.asarray(x["best_chain"])[:n][g["read_pair"]]]) and not g[key][1::2].any()
            elif key == "read_mapq":
                assert np.allclose(g[key], e[key], rtol=1e-9, atol=1e-15)
            elif isinstance(e[key], np.ndarray):
                assert np.array_equal(g[key], e[key], equal_nan=True), key
            else:
                assert g[key] == e[key], key
        if mac == 1000:
            assert e["n_reads"] == 0 and e["n_pairs_broken"] > 0
        else:
            assert e["n_reads"] > 10 and (e["read_weighted_ok"][1::2] == -1).all()
            prm = pkg.default_filter_params(first20_n=4)
            ug, ig, sg = pkg.filter_positions(C.CDLL(pkg.LIB_PATH), g, prm)
            ue, ie, se = ob.filter_positions(e, prm)
            assert np.array_equal(ug, ue) and np.array_equal(ig, ie) and sg == se


@pytest.mark.parametrize("seed,G,k,n_reads,lo,hi", [(7, 30000, 1, 40, 2000, 9000), (8, 20000, 3, 30, 600, 4000)], ids=["long-k1", "long-k3"])
def test_long_reads_match_oracle(pkg, oracle, seed, G, k, n_reads, lo, hi):
    """BASELINE config 5 style: kilobase reads with indel-rich CIGARs (thousands of operations), max_columns = 16384:
    the slab-backed projection kernel, the multi-round log-likelihood sum and the unpaired selection."""
    w = synth.make_world(seed=seed, G=G, k=k)
    u = synth.make_long_batch(w, n_reads, seed=seed + 1, len_lo=lo, len_hi=hi)
    cols = 16384
    o = oracle(w["graph"], w["contigs"], insert_mean=200.0, insert_sd=35.0, rng_seed=3, long_read_mode=1, max_columns=cols)
    e = o.align_long_reads(u)
    ctx = pkg.Context(w["graph"], w["contigs"], insert_mean=200.0, insert_sd=35.0, rng_seed=3, long_read_mode=1, max_columns=cols)
    gb = ctx.batch_unpaired(u); gb.align()
    st = gb.stats()
    assert st.n_errors == 0 and st.n_dp_calls == 0
    compare_chains(gb.chains(0), e["seeds"], u["n_chains"], check_ll=False, check_dp=False, label="long seeds")
    compare_chains(gb.chains(1), e["ext"], u["n_chains"], check_dp=False, label="long padded chains")
    g = gb.pairs(); x = e["pairs"]; n = n_reads
    for key in PAIR_INT:
        per = {"pair_status": n, "best_chain": n, "n_combinations": n, "n_cols": n}.get(key, n * cols)
        assert np.array_equal(np.asarray(g[key])[:per], np.asarray(x[key])[:per]), key
    assert np.allclose(g["pair_ll"][:n], x["pair_ll"][:n], rtol=1e-12, atol=0)
    assert int(np.asarray(x["n_cols"])[:n].max()) > 2000 and int(np.diff(u["cigar_off"]).max()) > 64


def test_many_cigar_operations_in_the_512_column_variant(pkg, oracle):
    """Error-rich reads of a few hundred bases have more than 64 CIGAR operations; the LDS-resident projection variant (max_columns <= 512)
    walks them in rounds of 64 like the slab-backed one (it used to flag such records as invalid input)."""
    w = synth.make_world(seed=21, G=8000, k=1)
    u = synth.make_long_batch(w, 200, seed=22, len_lo=150, len_hi=300)
    assert int(np.diff(u["cigar_off"]).max()) > 64
    kw = dict(insert_mean=200.0, insert_sd=35.0, rng_seed=3, long_read_mode=1, max_columns=512)
    e = oracle(w["graph"], w["contigs"], **kw).align_long_reads(u)
    ctx = pkg.Context(w["graph"], w["contigs"], **kw)
    gb = ctx.batch_unpaired(u); gb.align()
    assert gb.stats().n_errors == 0
    compare_chains(gb.chains(0), e["seeds"], u["n_chains"], check_ll=False, check_dp=False, label="seeds, > 64 operations")
    compare_chains(gb.chains(1), e["ext"], u["n_chains"], check_dp=False, label="chains, > 64 operations")
    g = gb.pairs(); x = e["pairs"]; n = 200
    for key in PAIR_INT:
        per = {"pair_status": n, "best_chain": n, "n_combinations": n, "n_cols": n}.get(key, n * 512)
        assert np.array_equal(np.asarray(g[key])[:per], np.asarray(x[key])[:per]), key
