"""BASELINE config 5 at FULL size under pytest: 50 000 synthetic ONT-like reads of ~10 kb (5e8 bases) in long-read mode against a 1 M-level graph --
processBAM::alignOneLongRead (mapper/processBAM.cpp:3618-3838: projection, padding to the full read, scoring with the long-read rates, no
extension DP) and assignMappingQualities_unpaired (:3900-4059), rows of 16 384 columns.  Checked over every one of the ~5.6e8 output columns:
checkChainConcordanceWithSequence (verboseSeedChain.cpp:48-77), checkLevelContiguity (verboseSeedChain.h:282-315), columns against the graph, the
mapping qualities; the 125 copies of every distinct read come out identical wherever they sit in the batch; the first distinct reads equal the oracle
bit for bit."""
import numpy as np
import pytest

from tools import synth

pytestmark = pytest.mark.gpu

N_DISTINCT = 400
N_READS = 50_000


def _tile_off(o, k):
    return np.concatenate([[0]] + [np.asarray(o[1:], np.int64) + i * int(o[-1]) for i in range(k)]).astype(np.int64)


def test_fifty_thousand_ten_kilobase_reads(pkg, oracle):
    w = synth.make_world(seed=2, G=1_000_000, k=1, n_mut=3)
    u0 = synth.make_long_batch(w, N_DISTINCT, seed=5, len_lo=9000, len_hi=11000)
    rep = N_READS // N_DISTINCT
    u = dict(n_pairs=N_DISTINCT * rep, read_off=_tile_off(u0["read_off"], rep), read_bases=np.tile(u0["read_bases"], rep), read_quals=np.tile(u0["read_quals"], rep),
             chain_off=_tile_off(u0["chain_off"], rep), read_primary=np.concatenate([u0["read_primary"] + i * u0["n_chains"] for i in range(rep)]).astype(np.int32),
             n_chains=u0["n_chains"] * rep, chain_contig=np.tile(u0["chain_contig"], rep), chain_pos=np.tile(u0["chain_pos"], rep), chain_offset=np.tile(u0["chain_offset"], rep),
             chain_as=np.tile(u0["chain_as"], rep), chain_reverse=np.tile(u0["chain_reverse"], rep), cigar_off=_tile_off(u0["cigar_off"], rep), cigar=np.tile(u0["cigar"], rep))
    n = u["n_pairs"]
    assert n == N_READS and int(u["read_off"][-1]) > 4.5e8
    kw = dict(insert_mean=200.0, insert_sd=35.0, rng_seed=3, long_read_mode=1, max_columns=16384)
    ctx = pkg.Context(w["graph"], w["contigs"], **kw)
    gb = ctx.batch_unpaired(u); gb.align()
    st = gb.stats()
    assert st.n_errors == 0 and st.n_dp_calls == 0
    pk = gb.pairs_packed(); sc = gb.pairs_scalars()
    off = pk["col_off"]; ncols = np.diff(off)
    assert (sc["pair_status"] == 0).all() and np.all(ncols >= np.diff(u["read_off"]))
    read_of_col = np.repeat(np.arange(n, dtype=np.int32), ncols)
    # ---- chain concordance: the non-gap read characters of all columns are the reads, in order
    s = pk["col_schar"]; isbase = s != ord("_")
    assert isbase.sum() == u["read_bases"].size and np.array_equal(s[isbase], u["read_bases"])
    # ---- level contiguity, columns against the graph
    lv = pk["col_level"]; d = np.nonzero(lv != -1)[0]
    same = read_of_col[d[1:]] == read_of_col[d[:-1]]
    assert np.all((lv[d[1:]] - lv[d[:-1]])[same] == 1)
    g = w["graph"]; ed = pk["col_edge"]; gc = pk["col_gchar"]; has = ed >= 0
    assert np.array_equal(has, lv != -1)
    assert np.array_equal(g["node_level"][g["edge_from"][ed[has]]], lv[has]) and np.array_equal(g["edge_label"][ed[has]], gc[has])
    assert np.all(gc[~has] == ord("_")) and np.all(s[~has] != ord("_"))
    assert np.all(sc["pair_mapq"] > 0) and np.all(sc["pair_mapq"] <= 1 + 1e-12) and pk["col_mapq"].min() >= 33
    # ---- copies of a read are identical wherever they sit
    ncopy = ncols.reshape(rep, N_DISTINCT)
    assert np.all(ncopy == ncopy[0])
    blk = int(off[N_DISTINCT])                                  # columns of one block of distinct reads
    for k in ("col_level", "col_gchar", "col_schar", "col_mapq"):
        a = pk[k][:blk]
        for i in (1, rep // 2, rep - 1):
            assert np.array_equal(pk[k][i * blk:(i + 1) * blk], a), (k, i)
    assert np.array_equal(sc["pair_ll"].reshape(rep, N_DISTINCT)[rep - 1], sc["pair_ll"][:N_DISTINCT])
    # ---- the first distinct reads against the oracle
    m = 48
    c1 = int(u0["chain_off"][m]); b1 = int(u0["read_off"][m]); g1 = int(u0["cigar_off"][c1])
    sub = dict(n_pairs=m, read_off=u0["read_off"][:m + 1], read_bases=u0["read_bases"][:b1], read_quals=u0["read_quals"][:b1], chain_off=u0["chain_off"][:m + 1],
               read_primary=u0["read_primary"][:m], n_chains=c1, chain_contig=u0["chain_contig"][:c1], chain_pos=u0["chain_pos"][:c1], chain_offset=u0["chain_offset"][:c1],
               chain_as=u0["chain_as"][:c1], chain_reverse=u0["chain_reverse"][:c1], cigar_off=u0["cigar_off"][:c1 + 1], cigar=u0["cigar"][:g1])
    e = oracle(w["graph"], w["contigs"], **kw).align_long_reads(sub)["pairs"]
    stride = 16384
    for r in range(m):
        k0 = int(e["n_cols"][r])
        assert k0 == ncols[r]
        for key in ("col_level", "col_edge", "col_gchar", "col_schar", "col_mapq"):
            assert np.array_equal(np.asarray(e[key])[r * stride:r * stride + k0], pk[key][off[r]:off[r] + k0]), (r, key)
    assert np.allclose(sc["pair_ll"][:m], e["pair_ll"][:m], rtol=1e-12, atol=0) and np.array_equal(sc["best_chain"][:m], e["best_chain"][:m])
