"""BASELINE config 3 at size: a sample of 8 x 2^20 = 8.4 M synthetic 2x150 bp pairs (2.5e9 read bases: beyond 32-bit offsets) goes from BAM
bytes through ONE context via the C ABI -- decoded on all host threads into one sample with 64-bit offsets, cut into windows of 2^20 pairs without
copying, two batches in flight -- and every batch keeps the reference's invariants.  Per batch:
  * the decoded window equals the generator's pairs (bases, qualities, number / scores of the alignments) under the name order;
  * no chain and no pair flagged; checkChainConcordanceWithSequence (verboseSeedChain.cpp:48-77), checkLevelContiguity (verboseSeedChain.h:282-315) and
    the columns against the graph over every output column;
  * truth (simulator/trueReadLevels.cpp:18-196): >= 99 % of the read bases on the level they were drawn from -- oracle-independent;
  * sampled bit-exact parity: 3 x 64 pairs of the batch re-run alone with their absolute chain numbers equal their rows of the big run and the oracle, and (round 4)
    a block of 8 192 consecutive pairs of the big run itself equals the oracle run on all host cores -- 65 536 + 1 536 pairs of the sample.
Reference: processBAM::alignReads_and_inferHLA's walk over the sample (mapper/processBAM.cpp:1788-1923), extractSeeds2 (:703-864)."""
import ctypes as C
import os
import time

import numpy as np
import pytest

from tools import synth

pytestmark = pytest.mark.gpu

CHUNK = 1 << 20
N_CHUNKS = int(os.environ.get("HLALA_CONFIG3_CHUNKS", "8"))


def test_eight_million_pairs_stream_through_one_context(pkg, oracle, tmp_path):
    from hla_la_amd import dist as D
    lib = C.CDLL(pkg.LIB_PATH)
    t0 = time.time()
    w = synth.make_world_m(seed=2)
    nct = w["contigs"]["n_contigs"]; clen = np.diff(w["contigs"]["contig_off"])
    refs = [("ctg%d" % i, int(clen[i])) for i in range(nct)]
    intervals = [("ctg%d" % i, 0, int(clen[i]) - 1, i) for i in range(nct)]
    path = tmp_path / "sample.bam"
    bw = synth.BamWriter(path, refs, threads=0, level=1)
    ranks = []
    for k in range(N_CHUNKS):
        b = synth.make_batch_m(w, CHUNK, seed=1000 + k)
        names, rank = synth.scrambled_names(k, CHUNK)
        bw.append_batch(b, names, order="coordinate")
        ranks.append(rank); del b
    size = bw.close()
    t_gen = time.time() - t0
    # ---- decode: the whole sample, 64-bit offsets
    t0 = time.time()
    S = pkg.bam_open_seeds(lib, path, intervals, threads=0)
    t_dec = time.time() - t0
    n = N_CHUNKS * CHUNK
    assert S.n_units == n and S.counts["incomplete"] == 0
    if N_CHUNKS >= 8:
        assert S.window(n - 1, 1).read_off[2] == 300 * n > 2 ** 31
        with pytest.raises(pkg.HlalaError, match="more than one batch holds"):
            S.window(0, n)
    S.pin(True)
    kw = dict(insert_mean=200.0, insert_sd=35.0, rng_seed=12345, max_columns=384)
    ctx = pkg.Context(w["graph"], w["contigs"], **kw)
    g = w["graph"]
    t_align = 0.0; t_check = 0.0
    acc_all = []
    nxt = ctx.batch_window(S, 0, CHUNK); nxt.align()
    for k in range(N_CHUNKS):
        gb = nxt
        if k + 1 < N_CHUNKS:                                               # two batches in flight: the next window is uploaded and queued before this one is read
            nxt = ctx.batch_window(S, (k + 1) * CHUNK, CHUNK); nxt.align()
        t0 = time.time()
        st = gb.stats()
        t_align += time.time() - t0
        assert st.n_errors == 0, (k, st.n_errors)
        pk = gb.pairs_packed(); sc = gb.pairs_scalars()
        t1 = time.time()
        # ---- the window against the generator (names: chunk prefix + scrambled pair number, so window k is chunk k in rank order)
        b = synth.make_batch_m(w, CHUNK, seed=1000 + k)
        inv = np.argsort(ranks[k])                                          # unit u of the window = pair inv[u] of the chunk
        d = S.to_dict(k * CHUNK, CHUNK)
        rsel = (2 * inv[:, None] + np.arange(2)[None, :]).reshape(-1)       # reads of the window in generator numbering
        assert np.array_equal(np.diff(d["read_off"]), np.full(2 * CHUNK, 150))
        assert np.array_equal(d["read_bases"].reshape(-1, 150), b["read_bases"].reshape(-1, 150)[rsel]) and np.array_equal(d["read_quals"].reshape(-1, 150), b["read_quals"].reshape(-1, 150)[rsel])
        assert np.array_equal(np.diff(d["chain_off"]), np.diff(b["chain_off"])[rsel])
        as_sum = np.add.reduceat(b["chain_as"].astype(np.int64), b["chain_off"][:-1].astype(np.int64))
        assert np.array_equal(np.add.reduceat(d["chain_as"].astype(np.int64), d["chain_off"][:-1]), as_sum[rsel])
        assert np.array_equal(d["chain_as"][d["read_primary"]], b["chain_as"][b["read_primary"]][rsel]) and np.array_equal(d["chain_pos"][d["read_primary"]], b["chain_pos"][b["read_primary"]][rsel])
        assert d["first_chain"] == (0 if k == 0 else prev_end)
        prev_end = d["first_chain"] + d["n_chains"]
        # ---- invariants over every column
        off = pk["col_off"]; T = pk["n_cols_total"]
        assert (sc["pair_status"] == 0).all() and T > 300 * CHUNK
        ncols = np.diff(off)
        assert ncols.min() >= 150
        read_of_col = np.repeat(np.arange(2 * CHUNK, dtype=np.int32), ncols)
        s = pk["col_schar"]; isbase = s != ord("_")
        assert isbase.sum() == d["read_bases"].size and np.array_equal(s[isbase], d["read_bases"])                    # chain concordance
        lv = pk["col_level"]; dd = np.nonzero(lv != -1)[0]
        same = read_of_col[dd[1:]] == read_of_col[dd[:-1]]
        assert np.all((lv[dd[1:]] - lv[dd[:-1]])[same] == 1)                                                         # level contiguity
        ed = pk["col_edge"]; gc = pk["col_gchar"]; has = ed >= 0
        assert np.array_equal(has, lv != -1)
        assert np.array_equal(g["node_level"][g["edge_from"][ed[has]]], lv[has]) and np.array_equal(g["edge_label"][ed[has]], gc[has])
        assert np.all(gc[~has] == ord("_")) and np.all(s[~has] != ord("_"))
        assert np.all(sc["pair_mapq"] > 0) and np.all(sc["pair_mapq"] <= 1 + 1e-12) and pk["col_mapq"].min() >= 33
        # ---- truth
        tl = b["truth_level"].reshape(-1, 150)[rsel].reshape(-1)
        al = lv[isbase]; known = tl >= 0
        acc = float((al[known] == tl[known]).mean()); acc_all.append(acc)
        assert acc >= 0.99, (k, acc)
        # ---- sampled parity: re-run alone with the absolute chain numbers; equal the rows of the big run and the oracle
        for start in (0, CHUNK // 2 + 17 * k, CHUNK - 64):
            sub, p0, c0 = D.shard_pairs_range(d, start, start + 64)
            sub["insert_mean"], sub["insert_sd"] = kw["insert_mean"], kw["insert_sd"]
            first = d["first_chain"] + c0
            exp = oracle(w["graph"], w["contigs"], **dict(kw, rng_seed=(12345 + 2 * first) & 0xFFFFFFFF)).align_batch(sub)["pairs"]
            gs = ctx.batch(sub); gs.set_first_chain(first); gs.align(); got = gs.pairs()
            for key in ("pair_status", "n_combinations", "best_chain", "n_cols", "col_level", "col_edge", "col_gchar", "col_schar", "col_mapq"):
                assert np.array_equal(got[key], exp[key]), (k, start, key)
            assert np.allclose(got["pair_ll"], exp["pair_ll"], rtol=1e-12, atol=0)
            r0 = 2 * start
            for r in range(r0, r0 + 128, 5):
                m = int(ncols[r])
                assert m == got["n_cols"][r - r0] and np.array_equal(lv[off[r]:off[r] + m], got["col_level"][(r - r0) * 384:(r - r0) * 384 + m])
            assert np.array_equal(sc["best_chain"][r0:r0 + 128], got["best_chain"] + c0) and np.array_equal(sc["pair_ll"][start:start + 64], got["pair_ll"])
            gs.close()
        # ---- a block of 8 192 consecutive pairs of the BIG run itself against the oracle on all host cores (orc_align_batch_mt: pairs are independent; every DP
        # draws its random seed from its chain's absolute number): rows of the big run, not a re-run -- 65 536 pairs of the sample in all
        BLK = 8192; start = (CHUNK // 3 + 4099 * k) % (CHUNK - BLK)
        sub, p0, c0 = D.shard_pairs_range(d, start, start + BLK)
        sub["insert_mean"], sub["insert_sd"] = kw["insert_mean"], kw["insert_sd"]
        first = d["first_chain"] + c0
        exp = oracle(w["graph"], w["contigs"], **dict(kw, rng_seed=(12345 + 2 * first) & 0xFFFFFFFF)).align_batch_mt(sub, 0, pairs_only=True)["pairs"]
        r0 = 2 * start
        assert np.array_equal(sc["pair_status"][start:start + BLK], exp["pair_status"]) and np.array_equal(sc["n_combinations"][start:start + BLK], exp["n_combinations"])
        assert np.array_equal(sc["best_chain"][r0:r0 + 2 * BLK], exp["best_chain"] + c0) and np.array_equal(ncols[r0:r0 + 2 * BLK], exp["n_cols"])
        assert np.allclose(sc["pair_ll"][start:start + BLK], exp["pair_ll"], rtol=1e-12, atol=0) and np.allclose(sc["pair_mapq"][start:start + BLK], exp["pair_mapq"], rtol=1e-9, atol=1e-12)
        ecols = {key: np.asarray(exp[key]).reshape(2 * BLK, 384) for key in ("col_level", "col_gchar", "col_schar", "col_mapq")}
        sel = np.arange(384)[None, :] < exp["n_cols"][:, None]
        a0, a1 = int(off[r0]), int(off[r0 + 2 * BLK])
        for key in ("col_level", "col_gchar", "col_schar", "col_mapq"):
            assert np.array_equal(ecols[key][sel], pk[key][a0:a1]), (k, key)
        gb.close(); del pk, sc, b, d
        t_check += time.time() - t1
    tm = S.timing()
    print(f"\nconfig 3: {n} pairs, BAM {size / 1e9:.2f} GB; generation + BAM writing {t_gen:.1f} s; decode {t_dec:.2f} s on {tm['threads']} threads "
          f"({ {k2: round(v, 2) for k2, v in tm.items() if k2 != 'threads'} }) = {n / t_dec / 1e6:.2f} M pairs/s; alignment (waiting for results) {t_align:.2f} s; "
          f"checks {t_check:.1f} s; truth accuracy per batch {[round(a, 4) for a in acc_all]}")
    S.close()
