"""CPU: pin the oracle on the known-answer material the reference itself carries for this path, plus hand-derived cases.

The reference cannot be built in this image (Boost/BamTools absent), so these are the reference-held checks that exist:
  * Utilities::intervalsOverlap start-up asserts          HLA-LA.cpp:94-102
  * Phred round-trip table of assignMappingQualities       mapper/processBAM.cpp:4216-4239
  * glibc rand_r as used by randomNumber_nonCritical        Utilities.cpp:922-927
"""
import ctypes as C

import numpy as np
import pytest

import oracle_binding as ob


def test_intervals_overlap_reference_asserts(oracle):
    L = ob.lib()
    f = lambda *a: bool(L.orc_intervals_overlap(*a))
    # HLA-LA.cpp:94-102, verbatim argument lists
    assert not f(1, 10, 11, 20)
    assert not f(5, 11, 1, 4)
    assert f(5, 11, 8, 11)
    assert f(8, 11, 1, 9)
    assert f(8, 11, 9, 10)
    assert f(9, 10, 8, 11)
    assert f(1, 10, 2, 3)
    assert f(2, 3, 1, 10)


def _phred(p):
    L = ob.lib()
    p = np.asarray(p, np.float64); out = np.zeros(len(p), np.uint8)
    L.orc_phred(len(p), p.ctypes.data_as(ob.P.c_f64p), out.ctypes.data_as(ob.P.c_u8p), None, None)
    return out


def _pcorrect(q):
    L = ob.lib()
    q = np.asarray(q, np.uint8); out = np.zeros(len(q), np.float64)
    L.orc_phred(len(q), None, None, q.ctypes.data_as(ob.P.c_u8p), out.ctypes.data_as(ob.P.c_f64p))
    return out


def test_phred_round_trip_table(oracle):
    # test_phredConversion(p, tol): assert(abs(p - pBack) <= pBack)   (processBAM.cpp:4216-4239)
    ps = [0.0, 0.5, 0.6, 0.7, 0.8, 0.9, 0.9, 0.99, 0.999, 0.9999, 1.0]
    ph = _phred(ps)
    back = _pcorrect(ph)
    for p, b in zip(ps, back):
        assert abs(p - b) <= b
    # published constants: p = 1 -> char 255 (Utilities.cpp:185-193 cap), p = 0 -> '!' (33), Phred 6 is the first level >= 0.7
    assert ph[-1] == 255 and ph[0] == 33
    assert _pcorrect([33 + 5])[0] < 0.7 <= _pcorrect([33 + 6])[0]          # SURVEY.md a17 parity trap
    assert _pcorrect([0])[0] == -1                                          # Utilities.cpp:359-362


def _rand_r_py(seed):
    """glibc stdlib/rand_r.c restated (three LCG rounds, 11 + 10 + 10 bits)."""
    nxt = seed
    nxt = (nxt * 1103515245 + 12345) & 0xFFFFFFFF
    res = (nxt // 65536) % 2048
    nxt = (nxt * 1103515245 + 12345) & 0xFFFFFFFF
    res = (res << 10) ^ ((nxt // 65536) % 1024)
    nxt = (nxt * 1103515245 + 12345) & 0xFFFFFFFF
    res = (res << 10) ^ ((nxt // 65536) % 1024)
    return res, nxt


def test_rand_r_restatement_matches_libc(oracle):
    L = ob.lib()
    seeds = np.array([0, 1, 12345, 2**31 - 1, 2**32 - 1, 987654321], np.uint32)
    s2 = seeds.copy(); vals = np.zeros(len(seeds), np.int32)
    L.orc_rand_r(len(seeds), s2.ctypes.data_as(ob.P.c_u32p), vals.ctypes.data_as(ob.P.c_i32p))     # libc rand_r
    for s, v, n in zip(seeds, vals, s2):
        pv, pn = _rand_r_py(int(s))
        assert pv == int(v) and pn == int(n)


# ------------------------------------------------------------------ hand-derived DP cases

def _linear_graph(seq, extra_edges=()):
    """One node per level, one edge per level labelled seq[i]; extra_edges = [(level, label)] parallel edges."""
    Lv = len(seq) + 1
    ef = list(range(len(seq))); et = [i + 1 for i in ef]; lab = [ord(c) for c in seq]
    for lv, c in extra_edges:
        ef.append(lv); et.append(lv + 1); lab.append(ord(c))
    return dict(n_levels=Lv, n_nodes=Lv, n_edges=len(ef), node_level=np.arange(Lv, dtype=np.int32),
                edge_from=np.asarray(ef, np.int32), edge_to=np.asarray(et, np.int32), edge_label=np.asarray(lab, np.uint8))


def _seed(read, begin, end, lv0):
    """Seed chain covering read[begin..end] on edges lv0.. of a linear graph (edge index = level)."""
    n = end - begin + 1
    return dict(n_reads=1, read_off=np.array([0, len(read)], np.int32), read_bases=np.frombuffer(read.encode(), np.uint8),
                read_quals=np.full(len(read), 33 + 30, np.uint8), n_chains=1, chain_read=np.array([0], np.int32),
                chain_seq_begin=np.array([begin], np.int32), chain_seq_end=np.array([end], np.int32),
                chain_reverse=np.array([0], np.uint8), col_off=np.array([0, n], np.int32),
                col_level=np.arange(lv0, lv0 + n, dtype=np.int32), col_edge=np.arange(lv0, lv0 + n, dtype=np.int32),
                col_gchar=np.frombuffer(read[begin:end + 1].encode(), np.uint8), col_schar=np.frombuffer(read[begin:end + 1].encode(), np.uint8))


def test_dp_exact_match_extension_by_hand(oracle):
    # graph = ACGTACGTACGTACGTACGT, read = levels 4..15; seed covers read[3..8]; both ends extend by matches: +2 per base
    g = "ACGTACGTACGTACGTACGT"
    read = g[4:16]
    o = oracle(_linear_graph(g), None)
    r = o.extend_seeds(_seed(read, 3, 8, 7))
    assert r["status"][0] == 0 and r["n_cols"][0] == 12
    assert r["col_level"][:12].tolist() == list(range(4, 16))
    assert bytes(r["col_gchar"][:12]).decode() == read and bytes(r["col_schar"][:12]).decode() == read
    assert r["dp_score"][:2].tolist() == [6, 6]          # 3 bases * S_match(2) on each side (alignerBase.cpp:19)
    assert r["col_fromseed"][:12].tolist() == [0, 0, 0, 1, 1, 1, 1, 1, 1, 0, 0, 0]


def test_dp_prefers_sequence_complete_alignment_by_hand(oracle):
    # right clip of 3 bases whose LAST base mismatches: local maximum is 4 after two matches, but the
    # sequence-complete cell (score 2+2-5 = -1 >= -16) is preferred (extensionAligner.cpp:1381-1472)
    g = "ACGTACGTACGTACG" + "CCCCC"       # no 'A' downstream: a deletion cannot rescue the last base
    read = g[4:15] + "A"
    o = oracle(_linear_graph(g), None)
    r = o.extend_seeds(_seed(read, 0, 8, 4))
    n = r["n_cols"][0]
    assert n == 12 and r["col_level"][:12].tolist() == list(range(4, 16))
    assert r["dp_score"][1] == 2 + 2 - 5
    assert bytes(r["col_schar"][:12]).decode() == read and bytes(r["col_gchar"][:12]).decode() == g[4:16]


def test_dp_affine_graph_gap_beats_two_mismatches_by_hand(oracle):
    # 2 clipped bases that cannot match: two mismatches cost -10, an affine gap in the graph costs
    # (S_openGap + S_extendGap) + S_extendGap = -6 - 2 = -8 (extensionAligner.cpp:629, 646); both are >= -16, and the
    # sequence-complete cell with the best score wins -> two insertion columns with level -1
    g = "AAAAAAAAAAAAAAAAAAAA"
    read = "AAAAAACC"
    o = oracle(_linear_graph(g), None)
    r = o.extend_seeds(_seed(read, 0, 5, 4))
    assert r["dp_score"][1] == -8 and r["n_cols"][0] == 8
    assert r["col_level"][:8].tolist() == [4, 5, 6, 7, 8, 9, -1, -1]
    assert bytes(r["col_gchar"][:8]).decode() == "AAAAAA__" and bytes(r["col_schar"][:8]).decode() == read


def test_gap_paths_by_hand(oracle):
    # levels 0..6; '_' edges parallel to the base edges at levels 2,3,4 -> maximal gap run from node 2;
    # every node on the way has a non-gap edge, so paths 2->3, 2->4, 2->5 complete (Graph.cpp:347-476)
    g = _linear_graph("ACGTAC", extra_edges=[(2, "_"), (3, "_"), (4, "_")])
    o = oracle(g, None)
    first, last, length = o.graph_paths()
    assert sorted(zip(first.tolist(), last.tolist(), length.tolist())) == [(2, 3, 1), (2, 4, 2), (2, 5, 3)]
    assert o.graph_gap_stretch().tolist() == [0, 0, 1, 1, 1, 0]      # run of 3 levels with a '_' edge (processBAM.cpp:93)


def test_dp_jumps_over_gap_path_by_hand(oracle):
    # read skips levels 5..9 through '_' edges: the right extension must take the gap path at cost 0 and spell the read
    g = "ACGTACGTTTTTACGTACGT"
    gaps = [(i, "_") for i in range(8, 12)]
    read = g[2:8] + g[12:18]
    o = oracle(_linear_graph(g, extra_edges=gaps), None)
    r = o.extend_seeds(_seed(read, 0, 5, 2))
    n = r["n_cols"][0]
    lv = r["col_level"][:n].tolist(); s = bytes(r["col_schar"][:n]).decode(); gg = bytes(r["col_gchar"][:n]).decode()
    assert lv == list(range(2, 18)) and s == g[2:8] + "____" + g[12:18] and gg == g[2:8] + "____" + g[12:18]
    assert r["dp_score"][1] == 12
