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


def _sub_reads(u, idx):
    """The unpaired batch of reads `idx` (one chain per read) of batch u."""
    idx = np.asarray(idx)
    ro = np.asarray(u["read_off"], np.int64); co = np.asarray(u["cigar_off"], np.int64)
    rl = (ro[idx + 1] - ro[idx]); cl = (co[idx + 1] - co[idx])
    take_b = np.concatenate([np.arange(ro[i], ro[i + 1]) for i in idx]); take_c = np.concatenate([np.arange(co[i], co[i + 1]) for i in idx])
    n = len(idx); ar = np.arange(n + 1, dtype=np.int32)
    return dict(n_pairs=n, read_off=np.concatenate([[0], np.cumsum(rl)]).astype(np.int32), read_bases=u["read_bases"][take_b], read_quals=u["read_quals"][take_b],
                chain_off=ar, read_primary=ar[:-1].copy(), n_chains=n, chain_contig=u["chain_contig"][idx], chain_pos=u["chain_pos"][idx], chain_offset=u["chain_offset"][idx],
                chain_as=u["chain_as"][idx], chain_reverse=u["chain_reverse"][idx], cigar_off=np.concatenate([[0], np.cumsum(cl)]).astype(np.int32), cigar=u["cigar"][take_c])


def test_distinct_long_reads_on_graph_m(pkg, oracle):
    """Variety instead of copies: 6 000 DISTINCT reads over 6-14 kb of reference on a Graph M world (backbone haplotypes with gap stretches, gene
    windows with hundreds to thousands of allele paths).  4 000 lie anywhere on the backbone contigs; 1 600 are laid ACROSS a gene window (they enter the
    allele-rich levels from the backbone and leave them again: the projection's chunked and level-by-level forms, levels with hundreds of nodes); 400 come
    from allele contigs of the windows.  Every column is checked against the reference's invariants and the graph, and 2 048 reads -- 1 024 of them
    window-crossing or allele reads -- bit for bit against the oracle."""
    w = synth.make_world_m(seed=9, n_levels=400_000, n_windows=6, alleles=(400, 3000))
    c = w["contigs"]; off = np.asarray(c["contig_off"], np.int64); clen = np.diff(off); cw = np.asarray(w["contig_window"])
    backbone = np.nonzero(cw < 0)[0]; allele = np.nonzero(cw >= 0)[0]
    rng = np.random.default_rng(77)
    starts = []
    for _ in range(4000):                                    # anywhere on the backbone
        h = int(backbone[rng.integers(0, len(backbone))]); starts.append((h, int(rng.integers(0, clen[h] - 14010))))
    wf = w["windows"]["first_level"]; wl = w["windows"]["last_level"]
    for i in range(1600):                                    # across a gene window: the read starts 1-5 kb in front of it
        k = i % len(wf); h = int(backbone[rng.integers(0, len(backbone))])
        lv = c["contig_level"][off[h]:off[h + 1]]
        p = int(np.searchsorted(lv, wf[k])) - int(rng.integers(1000, 5000))
        starts.append((h, max(0, min(p, int(clen[h]) - 14010))))
    for _ in range(400):                                     # allele contigs (3-6 kb: the whole contig or its tail)
        h = int(allele[rng.integers(0, len(allele))]); starts.append((h, int(rng.integers(0, max(1, clen[h] // 4)))))
    bs = synth.make_long_batches_parallel(w, len(starts), per_batch=750, seed=900, len_lo=6000, len_hi=14000, procs=8, starts=starts)
    # one batch: concatenate the chunks
    def cat_off(key):
        out = [np.zeros(1, np.int64)]
        for b in bs:
            out.append(np.asarray(b[key][1:], np.int64) + out[-1][-1])
        return np.concatenate(out)
    n = sum(b["n_pairs"] for b in bs); ar = np.arange(n + 1, dtype=np.int64)
    u = dict(n_pairs=n, n_chains=n, read_off=cat_off("read_off"), cigar_off=cat_off("cigar_off"), chain_off=ar, read_primary=ar[:-1].astype(np.int32))
    for k in ("read_bases", "read_quals", "chain_contig", "chain_pos", "chain_offset", "chain_as", "chain_reverse", "cigar"):
        u[k] = np.concatenate([b[k] for b in bs])
    assert n == 6000 and len(np.unique(u["chain_pos"].astype(np.int64) * 1000 + u["chain_contig"])) > 5900          # distinct reads
    crosses = np.zeros(n, bool); crosses[4000:] = True
    kw = dict(insert_mean=200.0, insert_sd=35.0, rng_seed=3, long_read_mode=1, max_columns=16384)
    ctx = pkg.Context(w["graph"], w["contigs"], **kw)
    gb = ctx.batch_unpaired(u); gb.align()
    st = gb.stats()
    assert st.n_dp_calls == 0
    pk = gb.pairs_packed(); sc = gb.pairs_scalars()
    okr = sc["pair_status"] == 0
    assert okr.mean() > 0.995, okr.mean()                    # (a read may exceed the 16 384-column row when it crosses a long gap stretch: flagged, not dropped silently)
    off_c = pk["col_off"]; ncols = np.diff(off_c)
    assert np.all(ncols[okr] >= np.diff(u["read_off"])[okr])
    read_of_col = np.repeat(np.arange(n, dtype=np.int32), ncols)
    s = pk["col_schar"]; isbase = s != ord("_")
    # chain concordance of every aligned read
    nb = np.bincount(read_of_col[isbase], minlength=n)
    assert np.array_equal(nb[okr], np.diff(u["read_off"])[okr])
    keep_b = np.repeat(okr, np.diff(u["read_off"]))
    assert np.array_equal(s[isbase], u["read_bases"][keep_b])
    # level contiguity, columns against the graph
    lv = pk["col_level"]; d = np.nonzero(lv != -1)[0]
    same = read_of_col[d[1:]] == read_of_col[d[:-1]]
    assert np.all((lv[d[1:]] - lv[d[:-1]])[same] == 1)
    g = w["graph"]; ed = pk["col_edge"]; gc = pk["col_gchar"]; has = ed >= 0
    assert np.array_equal(has, lv != -1)
    assert np.array_equal(g["node_level"][g["edge_from"][ed[has]]], lv[has]) and np.array_equal(g["edge_label"][ed[has]], gc[has])
    # the window-crossing reads did walk allele-rich levels
    npl = w["nodes_per_level"]
    rich = np.zeros(n, bool); rich[np.unique(read_of_col[has][npl[lv[has]] >= 50])] = True
    assert rich[4000:5600].mean() > 0.9 and rich.sum() > 1500
    # ---- 2 048 reads against the oracle (round 6; 512 before): 1 024 from the backbone, 1 024 that cross a window or come from an allele contig
    pick = np.concatenate([np.arange(0, 4000, 3)[:1024], np.arange(4000, 6000, 1)[:1024]])
    sub = _sub_reads(u, pick)
    e = oracle(w["graph"], w["contigs"], **kw).align_long_reads(sub)["pairs"]
    stride = 16384
    for i, r in enumerate(pick):
        assert int(e["pair_status"][i]) == int(sc["pair_status"][r]), (i, r)
        if sc["pair_status"][r] != 0:
            continue
        k0 = int(e["n_cols"][i])
        assert k0 == ncols[r], (i, r)
        for key in ("col_level", "col_edge", "col_gchar", "col_schar", "col_mapq"):
            assert np.array_equal(np.asarray(e[key])[i * stride:i * stride + k0], pk[key][off_c[r]:off_c[r] + k0]), (r, key)
    assert np.allclose(sc["pair_ll"][pick], e["pair_ll"][:len(pick)], rtol=1e-12, atol=0) and np.array_equal(sc["best_chain"][pick] - pick, e["best_chain"][:len(pick)] - np.arange(len(pick)))


def _long_world_and_reads(n_back=300, n_cross=500):
    w = synth.make_world_m(seed=11, n_levels=200_000, n_windows=4, alleles=(200, 900))
    c = w["contigs"]; off = np.asarray(c["contig_off"], np.int64); clen = np.diff(off); cw = np.asarray(w["contig_window"])
    backbone = np.nonzero(cw < 0)[0]
    rng = np.random.default_rng(5)
    starts = []
    for _ in range(n_back):
        h = int(backbone[rng.integers(0, len(backbone))]); starts.append((h, int(rng.integers(0, clen[h] - 14010))))
    wf = w["windows"]["first_level"]
    for i in range(n_cross):
        k = i % len(wf); h = int(backbone[rng.integers(0, len(backbone))])
        lv = c["contig_level"][off[h]:off[h + 1]]
        p = int(np.searchsorted(lv, wf[k])) - int(rng.integers(500, 9000))
        starts.append((h, max(0, min(p, int(clen[h]) - 14010))))
    bs = synth.make_long_batches_parallel(w, len(starts), per_batch=len(starts), seed=901, len_lo=6000, len_hi=14000, procs=8, starts=starts)
    return w, bs[0]


def test_forms_of_the_long_read_projection_agree(pkg, monkeypatch):
    """The re-threading DP of a long read runs in one of several forms (kernel_project.hip): segments one per lane; long segments level by level in chunks of up to 62
    levels / 416 nodes, a level beyond that node by node; the whole read level by level when it has more long segments than its list holds or its window is not staged.
    The switches HLALA_LONG_CHUNK_NODES / HLALA_LONG_MAXSEGS / HLALA_LONG_ORDER move reads between the forms; the results must not move.  (The default form against
    the oracle: test_distinct_long_reads_on_graph_m.)"""
    w, u = _long_world_and_reads()
    kw = dict(insert_mean=200.0, insert_sd=35.0, rng_seed=3, long_read_mode=1, max_columns=16384)
    def run(env):
        for k in ("HLALA_LONG_CHUNK_NODES", "HLALA_LONG_MAXSEGS", "HLALA_LONG_ORDER"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        ctx = pkg.Context(w["graph"], w["contigs"], **kw)
        gb = ctx.batch_unpaired(u); gb.align()
        st = gb.stats(); pk = gb.pairs_packed(); sc = gb.pairs_scalars()
        return st, pk, sc
    st0, pk0, sc0 = run({})
    assert (sc0["pair_status"] == 0).mean() > 0.99
    for env in ({"HLALA_LONG_MAXSEGS": "0"}, {"HLALA_LONG_CHUNK_NODES": "8"}, {"HLALA_LONG_CHUNK_NODES": "1", "HLALA_LONG_MAXSEGS": "1"}, {"HLALA_LONG_ORDER": "0"}):
        st, pk, sc = run(env)
        assert np.array_equal(sc["pair_status"], sc0["pair_status"]), env
        for k in ("col_off", "col_level", "col_edge", "col_gchar", "col_schar", "col_mapq"):
            assert np.array_equal(pk[k], pk0[k]), (env, k)
        assert np.array_equal(sc["pair_ll"], sc0["pair_ll"]) and np.array_equal(sc["best_chain"], sc0["best_chain"]), env
        assert st.n_edges_touched == st0.n_edges_touched and st.n_seed_columns == st0.n_seed_columns, env
