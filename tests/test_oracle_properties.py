"""CPU: protocol and invariant tests of the oracle, modelled on the reference's self-tests
(`--action testChainExtension`, HLA-LA.cpp:1733-1861, and the paranoid asserts listed in SURVEY.md section 4)."""
import numpy as np
import pytest

from tools import synth
from util import chain_cols


def _clip_batch(w, n_pairs, seed, clip=10):
    """True placements with exactly `clip` soft-clipped bases at both ends: removeSequenceCharacters(.., 10) (HLA-LA.cpp:1805-1806)."""
    b = synth.make_batch(w, n_pairs, seed=seed, p_secondary=0.0, indel_read_frac=0.0, p_no_clip=1.0)
    L = 150
    cig = np.array([(clip << 4) | synth.OP["S"], ((L - 2 * clip) << 4) | synth.OP["M"], (clip << 4) | synth.OP["S"]] * b["n_chains"], np.uint32)
    b["cigar"] = cig
    b["cigar_off"] = (np.arange(b["n_chains"] + 1) * 3).astype(np.int32)
    b["chain_pos"] = (b["chain_pos"] + clip).astype(np.int32)
    return b


@pytest.mark.parametrize("k", [0, 1, 3])
def test_chain_extension_protocol_respells_read(oracle, k):
    w = synth.make_world(seed=21 + k, G=6000, k=k)
    b = _clip_batch(w, 150, seed=5 + k)
    r = oracle(w["graph"], w["contigs"], insert_mean=b["insert_mean"], insert_sd=b["insert_sd"]).align_batch(b)
    ext = r["ext"]
    total = matched = 0
    for c in range(b["n_chains"]):
        assert ext["status"][c] == 0
        lv, ed, g, s, fs = chain_cols(ext, c)
        rd = b["chain_off"].searchsorted(c, side="right") - 1
        read = bytes(b["read_bases"][b["read_off"][rd]:b["read_off"][rd + 1]])
        assert s.replace(b"_", b"") == read                      # assert(extendedSeed_sequence_noGaps == originalSequence), HLA-LA.cpp:1824-1835
        assert ext["seq_begin"][c] == 0 and ext["seq_end"][c] == len(read) - 1
        d = lv[lv != -1]
        assert np.all(np.diff(d) == 1)                            # verboseSeedChain::checkLevelContiguity
        total += len(g); matched += sum(1 for x, y in zip(g, s) if x == y)
    assert matched / total > 0.9                                  # "Quality: Sequence" line of the reference test


def test_pair_level_invariants(oracle):
    w = synth.make_world(seed=31, G=8000, k=1)
    b = synth.make_batch(w, 300, seed=32)
    r = oracle(w["graph"], w["contigs"], insert_mean=b["insert_mean"], insert_sd=b["insert_sd"]).align_batch(b)
    p, e = r["pairs"], r["ext"]
    assert np.all(p["pair_status"] == 0)
    assert np.all((p["pair_mapq"] > 0) & (p["pair_mapq"] <= 1))
    assert np.all(p["mate_mapq"] + 1e-12 >= np.repeat(p["pair_mapq"], 2))           # assert(mapQ_chain1 >= mapQ), processBAM.cpp:4103
    assert np.all(e["ll"][e["status"] == 0] <= 0)                                      # assert((l >= 0) && (l <= 1)), extensionAligner.cpp:180
    st = p["_stride"]
    for rd in range(2 * b["n_pairs"]):
        n = p["n_cols"][rd]
        q = p["col_mapq"][rd * st: rd * st + n]
        assert n > 0 and np.all(q >= 33)
        if p["n_combinations"][rd // 2] == 1:
            assert np.all(q == 255)                                                    # single combination: Phred of p = 1
    # chosen chain is one of the read's own extended chains
    for rd in range(2 * b["n_pairs"]):
        c = p["best_chain"][rd]
        assert b["chain_off"][rd] <= c < b["chain_off"][rd + 1] and e["status"][c] == 0


def test_duplicate_and_strand_filters(oracle):
    w = synth.make_world(seed=41, G=5000, k=1, extra_identical=2)      # identical haplotypes -> identical start//stop ids
    b = synth.make_batch(w, 200, seed=42, p_secondary=1.0, max_secondary=4)
    # flip the strand of some secondaries
    rng = np.random.default_rng(0)
    prim = set(b["read_primary"].tolist())
    for c in range(b["n_chains"]):
        if c not in prim and rng.random() < 0.2:
            b["chain_reverse"][c] ^= 1
    r = oracle(w["graph"], w["contigs"], insert_mean=b["insert_mean"], insert_sd=b["insert_sd"]).align_batch(b)
    st = r["ext"]["status"]
    assert (st == 1).sum() > 0 and (st == 2).sum() > 0 and (st < 0).sum() == 0
    for rd in range(2 * b["n_pairs"]):
        c0, c1 = b["chain_off"][rd], b["chain_off"][rd + 1]
        pr = b["read_primary"][rd]
        for c in range(c0, c1):
            if b["chain_reverse"][c] != b["chain_reverse"][pr]:
                assert st[c] == 1                                                      # processBAM.cpp:3216
        assert st[pr] in (0, 2)
