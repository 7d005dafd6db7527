"""Parity sweep: N random worlds and batches -- stand-in graphs with random haplotype / mutation / gap parameters and small Graph M worlds (allele-rich gene
windows) -- product against oracle: extended chains column by column, pair records, work counters.  tools/parity_sweep.py N [pairs] [first seed];
prints one line per world and PARITY SWEEP OK n/N (worlds whose generator produced a record the reference asserts on are skipped and named)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from tools import synth
from conftest import load_package
from oracle_binding import Oracle, OracleError
from util import compare_chains
from test_gpu_align import assert_pairs_equal


def sweep_world(P, s, n):
    """One random world of the sweep (seed s, n pairs): product against oracle; returns the batch statistics, None if the generator produced a record the
    reference asserts on (the oracle raises)."""
    rng = np.random.default_rng(s)
    t0 = time.time()
    read_len = int(rng.choice([76, 100, 125, 150, 151, 250]))
    if s % 3 == 2:
        al_lo = int(rng.choice([50, 400, 1500])); al_hi = al_lo + int(rng.choice([100, 1000, 3000]))
        nwin = int(rng.integers(1, 4))
        w = synth.make_world_m(seed=s, n_levels=int(rng.integers(30_000, 90_000)), n_windows=nwin, alleles=(al_lo, al_hi),
                               n_backbone=int(rng.integers(2, 9)), backbone_div=float(rng.choice([0.001, 0.003, 0.01])), gap_stretch_frac=float(rng.choice([0.0, 0.02, 0.08])))
        fg = float(rng.choice([0.1, 0.5, 1.0]))
        b = synth.make_batch_m(w, n, seed=s + 1, read_len=read_len, jump_mean=float(read_len + rng.integers(120, 300)), jump_sd=float(rng.integers(15, 60)),
                               clip_max=int(rng.integers(0, read_len // 2)), frac_gene=fg, p_secondary=float(rng.random()), max_secondary=int(rng.integers(1, 7)), p_random_secondary=float(rng.random() * 0.3))
        what = "graph M: %d windows, %d-%d alleles, gene share %.1f" % (nwin, al_lo, al_hi, fg)
    else:
        k = int(rng.choice([0, 1, 2, 3, 5])); G = int(rng.integers(8_000, 45_000))
        kw = dict(n_mut=int(rng.integers(2, 9)))
        if rng.random() < 0.4: kw["mut_density"] = float(rng.choice([0.01, 0.04, 0.08]))
        if rng.random() < 0.4: kw["n_largegap"] = int(rng.integers(1, 4))
        if rng.random() < 0.3: kw["gap_frac"] = float(rng.choice([0.2, 0.6]))
        if rng.random() < 0.3: kw["extra_identical"] = int(rng.integers(1, 3))
        w = synth.make_world(seed=s, G=G, k=k, **kw)
        read_len = max(read_len, 100)                      # (the stand-in generator places its clips 40 bases from the ends)
        b = synth.make_batch(w, n, seed=s + 1, read_len=read_len, ins_mean=float(read_len + rng.integers(20, 200)), ins_sd=float(rng.integers(10, 70)), clip_max=int(rng.integers(0, read_len // 2)),
                             p_secondary=float(rng.random()), max_secondary=int(rng.integers(1, 7)), p_random_secondary=float(rng.random() * 0.4), indel_read_frac=float(rng.choice([0.0, 0.05, 0.3])),
                             p_no_clip=float(rng.random() * 0.3))
        what = "stand-in: G %d, k %d, %s" % (G, k, kw)
    kwc = dict(insert_mean=b["insert_mean"], insert_sd=b["insert_sd"], rng_seed=s, max_columns=384)
    try:
        exp = Oracle(w["graph"], w["contigs"], **kwc).align_batch(b)
    except OracleError as err:
        print("seed %d skipped (%s): %s" % (s, what, str(err)[:120]), flush=True); return None
    ctx = P.Context(w["graph"], w["contigs"], **kwc)
    gb = ctx.batch(b); gb.align()
    compare_chains(gb.chains(1), exp["ext"], b["n_chains"], label="sweep %d" % s)
    # a pair with a chain of more than max_columns columns is FLAGGED by both (pair_status -1: the product's capacity, include/hlala_gpu.h) -- the oracle still fills the
    # pair's record with what the reference, which has no such limit, would return: compared only as far as the flag
    got = gb.pairs(); ep = {k: np.array(v, copy=True) for k, v in exp["pairs"].items() if isinstance(v, np.ndarray)}
    assert np.array_equal(got["pair_status"], ep["pair_status"])
    flagged = ep["pair_status"] != 0
    if flagged.any():
        npairs = len(flagged); fr = np.repeat(flagged, 2)
        for k, v in ep.items():
            if k not in got: continue
            if len(v) == npairs: v[flagged] = got[k][flagged]
            elif len(v) == 2 * npairs: v[fr] = got[k][fr]
            elif len(v) % (2 * npairs) == 0: v.reshape(2 * npairs, -1)[fr] = got[k].reshape(2 * npairs, -1)[fr]
    assert_pairs_equal(got, ep)
    st = gb.stats()
    assert (st.n_dp_calls, st.n_dp_iterations, st.n_dp_cells) == tuple(int(x) for x in exp["stats"][:3]), s
    extra = ""
    if s % 3 != 2 and s % 2 == 0:
        # the same reads once more as UNPAIRED reads in long-read mode (alignOneLongRead, mapper/processBAM.cpp:3618-3838: projection, padding, scoring, selection; no extension DP)
        u = synth.as_unpaired(b); nu = u["n_pairs"]
        ko = dict(insert_mean=200.0, insert_sd=35.0, rng_seed=s, long_read_mode=1)
        eo = Oracle(w["graph"], w["contigs"], **ko); e = eo.align_long_reads(u)
        cu = P.Context(w["graph"], w["contigs"], **ko)
        gu = cu.batch_unpaired(u); gu.align()
        assert gu.stats().n_dp_calls == 0
        compare_chains(gu.chains(0), e["seeds"], u["n_chains"], check_ll=False, check_dp=False, label="sweep %d unpaired seeds" % s)
        compare_chains(gu.chains(1), e["ext"], u["n_chains"], check_dp=False, label="sweep %d unpaired padded chains" % s)
        g = gu.pairs(); x = e["pairs"]; stride = eo.max_columns
        for key in ("pair_status", "best_chain", "n_combinations", "n_cols", "col_level", "col_edge", "col_gchar", "col_schar", "col_fromseed", "col_mapq"):
            per = {"pair_status": nu, "best_chain": nu, "n_combinations": nu, "n_cols": nu}.get(key, nu * stride)
            assert np.array_equal(np.asarray(g[key])[:per], np.asarray(x[key])[:per]), (s, key)
        assert np.allclose(g["pair_ll"][:nu], x["pair_ll"][:nu], rtol=1e-12, atol=0)
        extra = "; + %d unpaired reads in long-read mode" % nu
        gu.close(); cu.close()
    print("seed %d ok (%s; reads of %d): %d pairs, %d chains, %d DP calls by class %s, flagged %d%s, %.0f s" % (s, what, read_len, n, b["n_chains"], st.n_dp_calls, list(st.n_dp_class), st.n_errors, extra, time.time() - t0), flush=True)
    gb.close(); ctx.close()
    return st


if __name__ == "__main__":
    P = load_package()
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
    s0 = int(sys.argv[3]) if len(sys.argv) > 3 else 9000
    ok = 0; skipped = []
    for s in range(s0, s0 + N):
        if sweep_world(P, s, n) is None: skipped.append(s)
        else: ok += 1
    print("PARITY SWEEP OK %d/%d worlds bit-exact (chains, pairs, work counters); skipped %s" % (ok, N, skipped))
