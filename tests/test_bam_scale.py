"""The BAM decoder at the size of a sample (BASELINE config 3): 64-bit offsets, parallel inflate / parse / grouping / name sort.

  * the result does not depend on the number of decoding threads nor on where the rounds of the decoder cut the file (records that straddle
    rounds), and equals the source batch; a file written by the pure-Python writer of test_bam.py decodes to the same sample as the file
    written by the C++ writer (two independent encoders, one decoder);
  * a sample of more than 2^31 read bases (the size at which 32-bit offsets wrapped: 7.16 M pairs of 2x150 bp) decodes with correct 64-bit
    offsets; asking for the whole sample as ONE batch fails loudly, windows of it are handed out without copying.
Reference: processBAM::extractSeeds2 (mapper/processBAM.cpp:703-864), the read-name order of completeProtoSeeds (:2024-2039)."""
import ctypes as C
import os

import numpy as np
import pytest

from tools import synth


def _world(n_pairs, seed=4):
    w = synth.make_world(seed=seed, G=5000, k=1)
    b = synth.make_batch(w, n_pairs, seed=seed + 1)
    clen = np.diff(w["contigs"]["contig_off"]); nct = w["contigs"]["n_contigs"]
    refs = [("hap%d" % i, int(clen[i])) for i in range(nct)]
    intervals = [("hap%d" % i, 0, int(clen[i]) - 1, i) for i in range(nct)]
    return w, b, refs, intervals


def _same(a, b):
    for k in ("n_pairs", "n_chains"):
        assert a[k] == b[k], k
    for k in ("read_off", "read_bases", "read_quals", "chain_off", "read_primary", "chain_contig", "chain_pos", "chain_offset", "chain_as", "chain_reverse", "cigar_off", "cigar"):
        assert np.array_equal(a[k], b[k]), k


def test_decoding_does_not_depend_on_threads_rounds_or_the_writer(pkg, tmp_path, monkeypatch):
    from test_bam import batch_records, canon, write_bam
    lib = C.CDLL(pkg.LIB_PATH)
    n = 3000
    w, b, refs, intervals = _world(n)
    names, rank = synth.scrambled_names(3, n)
    path = tmp_path / "s.bam"
    bw = synth.BamWriter(path, refs, threads=3); bw.append_batch(b, names, order="coordinate"); assert bw.close() > 0
    one, nm1, cnt1 = pkg.bam_extract_seeds(lib, path, intervals, threads=1)
    assert one["n_pairs"] == n and cnt1["incomplete"] == 0 and cnt1["examined"] == b["n_chains"]
    # unit u of the sample is the pair whose scrambled name has rank u
    inv = np.argsort(rank)
    assert nm1 == [bytes(names[p]).decode() for p in inv]
    got = canon(one, 2 * n); src = canon(b, 2 * n)
    for u in range(n):
        p = int(inv[u])
        assert got[2 * u] == src[2 * p] and got[2 * u + 1] == src[2 * p + 1], u
    many, nm5, cnt5 = pkg.bam_extract_seeds(lib, path, intervals, threads=5)
    _same(one, many); assert nm5 == nm1 and cnt5 == cnt1
    monkeypatch.setenv("HLALA_BAM_SEGMENT_BYTES", "70000")                 # about one BGZF block per round: most records straddle rounds
    cut, nmc, cntc = pkg.bam_extract_seeds(lib, path, intervals, threads=4)
    _same(one, cut); assert nmc == nm1 and cntc == cnt1
    monkeypatch.delenv("HLALA_BAM_SEGMENT_BYTES")
    # the independent Python writer (every record carries its read; records in random order): the same units, the alignments of a mate as the
    # same multiset (equal scores may come out in another order when the file order differs -- std::sort, processBAM.cpp:1952-1961)
    recs = batch_records(b, np.random.default_rng(1))
    for r in recs:
        p = int(r["name"][4:]); r["name"] = bytes(names[p]).decode()
    path2 = tmp_path / "py.bam"
    write_bam(path2, refs, recs, block=20000)
    py, nmp, cntp = pkg.bam_extract_seeds(lib, path2, intervals, threads=3)
    assert nmp == nm1 and cntp == cnt1 and canon(py, 2 * n) == got


def test_windows_of_a_sample_are_its_batches(pkg, tmp_path):
    lib = C.CDLL(pkg.LIB_PATH)
    n = 500
    w, b, refs, intervals = _world(n, seed=8)
    names, rank = synth.scrambled_names(0, n)
    path = tmp_path / "s.bam"
    bw = synth.BamWriter(path, refs, threads=2); bw.append_batch(b, names, order="random", rng=np.random.default_rng(3)); bw.close()
    S = pkg.bam_open_seeds(lib, path, intervals, threads=2)
    assert S.n_units == n and set(S.timing()) == {"index", "inflate", "parse", "group", "name_sort", "layout", "threads"}
    whole = S.to_dict()
    a = S.to_dict(0, 200); z = S.to_dict(200, 300)
    assert a["first_chain"] == 0 and z["first_chain"] == int(whole["chain_off"][400]) and a["n_chains"] + z["n_chains"] == whole["n_chains"]
    assert np.array_equal(np.concatenate([a["read_bases"], z["read_bases"]]), whole["read_bases"]) and np.array_equal(np.concatenate([a["cigar"], z["cigar"]]), whole["cigar"])
    assert np.array_equal(z["read_primary"] + z["first_chain"], whole["read_primary"][400:]) and S.names(200, 2) == S.names()[200:202]
    with pytest.raises(pkg.HlalaError):
        S.window(400, 200)
    S.close()


def test_a_sample_beyond_2_31_bases_keeps_64_bit_offsets(pkg, tmp_path):
    """53 000 pairs of 2 x 20 300 bases = 2.15e9 read bases (> 2^31): the offsets of the decoded sample are exact, the whole sample is refused as ONE
    batch with a message, its windows are fine.  (Paired records of any length decode; the 1024-base limit of the paired DP is hlala_batch_create's.)"""
    lib = C.CDLL(pkg.LIB_PATH)
    L = 20300; n = 53000; per = 6625
    refs = [("c0", 50_000_000)]; intervals = [("c0", 0, 49_999_999, 0)]
    path = tmp_path / "big.bam"
    bw = synth.BamWriter(path, refs, threads=0, level=1)
    acgt = np.frombuffer(b"ACGT", np.uint8)
    base_rows = np.stack([acgt[(np.arange(L) + r) % 4] for r in range(4)])                       # read r of pair p: ACGT rotated by (2p + mate)
    qual_rows = np.stack([(33 + (np.arange(L) + r) % 40).astype(np.uint8) for r in range(40)])    # qualities 33 + ((p + i) % 40)
    for k in range(n // per):
        p = np.arange(k * per, (k + 1) * per, dtype=np.int64)
        nm = np.stack([np.frombuffer(b"0123456789abcdef", np.uint8)[(p >> sh) & 15] for sh in range(28, -4, -4)], axis=1)
        names = np.repeat(nm, 2, axis=0).reshape(-1); name_off = np.arange(2 * per + 1, dtype=np.int64) * 8
        flag = np.tile(np.array([1 | 64, 1 | 128 | 16], np.uint16), per)
        pos = np.repeat((p * 7) % 1_000_000, 2).astype(np.int32)
        cigar = np.full(2 * per, (L << 4) | 0, np.uint32); cigar_off = np.arange(2 * per + 1, dtype=np.int64)
        seq_off = np.arange(2 * per + 1, dtype=np.int64) * L
        rot = np.repeat(2 * p, 2) + np.tile(np.array([0, 1]), per)
        bases = base_rows[rot % 4].reshape(-1)
        quals = qual_rows[np.repeat(p, 2) % 40].reshape(-1)
        bw.append(names, name_off, flag, np.zeros(2 * per, np.int32), pos, cigar_off, cigar, seq_off, bases, quals, np.full(2 * per, 77, np.int32))
        del bases, quals
    assert bw.close() > 0
    S = pkg.bam_open_seeds(lib, path, intervals, threads=0)
    assert S.n_units == n
    total = 2 * n * L
    assert total > 2 ** 31
    with pytest.raises(pkg.HlalaError, match="more than one batch holds"):
        S.window(0, n)
    d = S.window(0, n // 2)                                   # half the sample: fine
    assert d.n_chains == n and d.read_off[2 * (n // 2)] == (n // 2) * 2 * L
    # the last units: offsets beyond 2^31, bases and names of the right pairs (names are hex of the pair index: name order = pair order)
    last = S.to_dict(n - 3, 3)
    d = S.window(n - 3, 3)
    assert d.read_off[0] == (n - 3) * 2 * L > 2 ** 31 and d.read_off[6] == total
    assert S.names(n - 3, 3) == ["%08x" % q for q in range(n - 3, n)]
    for j, q in enumerate(range(n - 3, n)):
        for m in range(2):
            r = 2 * j + m
            assert np.array_equal(last["read_bases"][r * L:(r + 1) * L], acgt[(np.arange(L) + 2 * q + m) % 4])
            assert np.array_equal(last["read_quals"][r * L:(r + 1) * L], (33 + (np.arange(L) + q) % 40).astype(np.uint8))
    assert np.array_equal(last["chain_reverse"], np.tile([0, 1], 3)) and np.all(last["chain_as"] == 77) and last["first_chain"] == 2 * (n - 3)
    S.close()
