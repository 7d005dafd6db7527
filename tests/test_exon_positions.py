"""Exon positions of read pairs for one locus (hla/HLATyper.cpp:1385-1428, 3192-3565, 3933-4083)."""
import numpy as np
import pytest

import oracle_binding as ob
from tools import synth


def locus(G, a, exons):
    """level_min = a; `exons` = list of (start offset, length) inside the locus; exon positions run through consecutively."""
    width = max(s + n for s, n in exons)
    l2e = np.full(width, -1, np.int32); k = 0
    for s, n in exons:
        l2e[s:s + n] = np.arange(k, k + n); k += n
    assert a + width < G
    return a, l2e


def fake_pairs(mates, stride=24):
    """mates = list of (levels, gchars, schars, mapq chars) per mate; reads are the non-gap sequence characters with quality chars given per base."""
    n = len(mates) // 2
    pr = dict(pair_status=np.zeros(n, np.int32), n_cols=np.zeros(2 * n, np.int32), col_level=np.zeros(2 * n * stride, np.int32),
              col_gchar=np.zeros(2 * n * stride, np.uint8), col_schar=np.zeros(2 * n * stride, np.uint8), col_mapq=np.zeros(2 * n * stride, np.uint8),
              mate_mapq=np.ones(2 * n), strands_valid=np.ones(n, np.uint8))
    off = [0]; bases = []; quals = []
    for r, (lv, g, s, mq, q) in enumerate(mates):
        m = len(lv); pr["n_cols"][r] = m
        pr["col_level"][r * stride: r * stride + m] = lv
        pr["col_gchar"][r * stride: r * stride + m] = np.frombuffer(g.encode(), np.uint8)
        pr["col_schar"][r * stride: r * stride + m] = np.frombuffer(s.encode(), np.uint8)
        pr["col_mapq"][r * stride: r * stride + m] = np.frombuffer(mq.encode(), np.uint8)
        rb = s.replace("_", ""); assert len(q) == len(rb)
        bases.append(np.frombuffer(rb.encode(), np.uint8)); quals.append(np.frombuffer(q.encode(), np.uint8)); off.append(off[-1] + len(rb))
    batch = dict(n_pairs=n, read_off=np.array(off, np.int32), read_bases=np.concatenate(bases), read_quals=np.concatenate(quals))
    return pr, batch, stride


def test_oracle_exon_positions_hand_derived(oracle):
    # mate 1: levels 10..14 with an insertion (level -1) after level 11 and a deletion in the read at level 13
    #   columns: (10,A,A) (11,C,C) (-1,_,G) (12,T,T) (13,G,_) (14,A,C)        read = A C G T C, qualities 5 6 7 8 9 (as chars)
    # mate 2: levels 13..16, all matches                                        read = G A T T, qualities I H G F
    m1 = ([10, 11, -1, 12, 13, 14], "AC_TGA", "ACGT_C", "JJJJJJ", "56789")
    m2 = ([13, 14, 15, 16], "GATT", "GATT", "KKKK", "IHGF")
    pr, batch, stride = fake_pairs([m1, m2])
    # locus: levels 11..15 are exon positions 0..4
    e = ob.exon_positions(pr, batch, stride, 11, np.arange(5), insert_mean=0, insert_sd=10)
    assert e["n_reads"] == 1 and e["n_pairs_ok"] == 1 and e["n_pairs_broken"] == 0
    assert e["read_pair"].tolist() == [0] and e["pos_off"].tolist() == [0, 5]
    # one position per level 11..15 (removeDoublePositionsFromRead): level 13 is "_" in mate 1 (Q = 0) and G/'I' in mate 2 -> mate 2;
    # level 14: mate 1 has C with quality '9' (57), mate 2 has A with 'H' (72) -> mate 2; level 11 carries the inserted G: genotype "CG", qualities "67"
    assert e["pos_level"].tolist() == [11, 12, 13, 14, 15] and e["pos_exon"].tolist() == [0, 1, 2, 3, 4]
    assert e["pos_mate"].tolist() == [1, 1, 2, 2, 2]
    geno = [bytes(e["geno_chars"][e["geno_off"][i]:e["geno_off"][i + 1]]).decode() for i in range(5)]
    qual = [bytes(e["qual_chars"][e["geno_off"][i]:e["geno_off"][i + 1]]).decode() for i in range(5)]
    assert geno == ["CG", "T", "G", "A", "T"] and qual == ["67", "8", "I", "H", "G"]
    assert e["pos_mapq"].tolist() == [ord("J"), ord("J"), ord("K"), ord("K"), ord("K")]
    # running novel gap: the insertion column and the deletion column are single gaps; level 11 sits next to the insertion (backward sweep 0,
    # forward 0 at its own column), level 12 is a two-character column (0)
    assert e["pos_novel_gap"].tolist() == [0, 0, 0, 0, 0]
    # weighted OK fractions: mate 1 has an insertion (1), a deletion (1) and a mismatch at level 14 weighted with pCorrect('9'); read length 5
    pc = ob.lib  # noqa
    import ctypes as C
    out = np.zeros(1); q = np.array([ord("9")], np.uint8)
    ob.lib().orc_phred(1, None, None, q.ctypes.data_as(C.POINTER(C.c_uint8)), out.ctypes.data_as(C.POINTER(C.c_double)))
    assert np.isclose(e["read_weighted_ok"][0], 1.0 - (2 + out[0]) / 5.0, rtol=1e-15) and e["read_weighted_ok"][1] == 1.0
    assert np.isclose(e["read_fraction_ok"][0], 3 / 6) and e["read_fraction_ok"][1] == 1.0
    # distance in graph levels: mate 1 first (10..14), mate 2 13..16 -> 13 - 14 - 1 = -2
    assert e["read_distance"].tolist() == [-2]
    # the pair test: distance too far from the insert mean -> broken, nothing reported
    e2 = ob.exon_positions(pr, batch, stride, 11, np.arange(5), insert_mean=500, insert_sd=10)
    assert e2["n_reads"] == 0 and e2["n_pairs_broken"] == 1
    # a read deletion followed by an insertion: "_" + inserted base -> the leading '_' is dropped (:3325-3333)
    m3 = ([20, 21, -1, 22], "ACG_T"[0:2] + "_" + "T", "A_" + "G" + "T", "JJJJ", "123")
    pr3, batch3, stride3 = fake_pairs([m3, ([30, 31], "AA", "AA", "JJ", "45")])
    e3 = ob.exon_positions(pr3, batch3, stride3, 20, np.arange(3), insert_mean=7, insert_sd=10)
    geno3 = [bytes(e3["geno_chars"][e3["geno_off"][i]:e3["geno_off"][i + 1]]).decode() for i in range(e3["n_pos"])]
    assert e3["pos_level"].tolist() == [20, 21, 22] and geno3 == ["A", "G", "T"]
    assert e3["pos_novel_gap"].tolist() == [0, 2, 0]


KEYS_INT = ("read_pair", "read_distance", "read_cols_nongap", "pos_off", "pos_exon", "pos_level", "pos_mate", "pos_mapq", "pos_novel_gap", "geno_off", "geno_chars", "qual_chars")


@pytest.mark.gpu
@pytest.mark.parametrize("seed,G,k,n_pairs", [(1, 5000, 1, 400), (3, 8000, 3, 300), (2, 8000, 0, 150)], ids=["seed1", "seed3", "seed2"])
def test_exon_positions_match_oracle(pkg, oracle, seed, G, k, n_pairs):
    w = synth.make_world(seed=seed, G=G, k=k)
    b = synth.make_batch(w, n_pairs, seed=seed + 10)
    o = oracle(w["graph"], w["contigs"], insert_mean=b["insert_mean"], insert_sd=b["insert_sd"], rng_seed=777)
    exp_pairs = o.align_batch(b)["pairs"]
    ctx = pkg.Context(w["graph"], w["contigs"], insert_mean=b["insert_mean"], insert_sd=b["insert_sd"], rng_seed=777)
    gb = ctx.batch(b); gb.align()
    total = 0
    for a, exons, kw in ((G // 4, [(0, 270), (400, 276), (900, 120)], {}), (G // 2, [(0, 800)], dict(min_mapq=0.5, min_weighted_ok=0.97)), (10, [(5, 40)], {})):
        lmin, l2e = locus(G, a, exons)
        mask = None
        if kw:
            mask = (np.arange(n_pairs) % 3 != 0).astype(np.uint8)         # as if includeInHLA had dropped a third of the pairs
        e = ob.exon_positions(exp_pairs, b, o.max_columns, lmin, l2e, b["insert_mean"], b["insert_sd"], pair_mask=mask, **kw)
        g = gb.exon_positions(lmin, l2e, b["insert_mean"], b["insert_sd"], pair_mask=mask, **kw)
        for key in ("n_reads", "n_pos", "n_chars", "n_pairs_ok", "n_pairs_broken"):
            assert g[key] == e[key], key
        for key in KEYS_INT:
            assert np.array_equal(g[key], e[key]), key
        # FP64 sums in column order with host-built pCorrect: bit-identical
        assert np.array_equal(g["read_weighted_ok"], e["read_weighted_ok"])
        assert np.array_equal(g["read_fraction_ok"], e["read_fraction_ok"], equal_nan=True)
        # oneExonPosition::mapQ of either mate (posteriors: device exp(), as in test_gpu_align) and ::reverse = strand of the chosen alignment
        assert np.allclose(g["read_mapq"], e["read_mapq"], rtol=1e-9, atol=1e-15)
        best = exp_pairs["best_chain"].reshape(-1, 2)[g["read_pair"]].reshape(-1)
        assert np.array_equal(g["read_reverse"], b["chain_reverse"][best])
        total += e["n_pos"]
    assert total > 1000
