"""BAM reader + seed extraction (mapper/processBAM.cpp:703-864, 1945-1967, 4314-4334; protoSeeds.cpp:23-36, 371-380).
The BAM file is written here following the SAM/BAM specification (BGZF blocks, little-endian records); the expected batch is derived
from the same record list with the reference's rules written out in Python."""
import os
import ctypes as C
import struct
import zlib

import numpy as np
import pytest

OPS = "MIDNSHP=X"
SEQ16 = "=ACMGRSVTWYHKDBN"
INT_FMT = {"c": "<b", "C": "<B", "s": "<h", "S": "<H", "i": "<i", "I": "<I"}


def bgzf_block(data):
    co = zlib.compressobj(6, zlib.DEFLATED, -15); comp = co.compress(data) + co.flush()
    bsize = len(comp) + 25
    return (struct.pack("<BBBBIBBH", 31, 139, 8, 4, 0, 0, 255, 6) + b"BC" + struct.pack("<HH", 2, bsize) + comp + struct.pack("<II", zlib.crc32(data) & 0xFFFFFFFF, len(data)))


def write_bam(path, refs, records, block=3000):
    """refs = [(name, length)]; records = dicts(name, flag, ref, pos, cigar=[(len, op)], seq, qual (Phred ints), tags = [(tag, type, value)])."""
    raw = b"BAM\x01" + struct.pack("<i", 0) + struct.pack("<i", len(refs))
    for nm, ln in refs:
        raw += struct.pack("<i", len(nm) + 1) + nm.encode() + b"\0" + struct.pack("<i", ln)
    for r in records:
        name = r["name"].encode() + b"\0"; cig = b"".join(struct.pack("<I", (l << 4) | OPS.index(o)) for l, o in r["cigar"])
        seq = r["seq"]; packed = bytearray((len(seq) + 1) // 2)
        for i, ch in enumerate(seq):
            packed[i // 2] |= SEQ16.index(ch) << (4 if i % 2 == 0 else 0)
        tags = b""
        for tag, ty, v in r.get("tags", []):
            tags += tag.encode() + ty.encode() + (struct.pack(INT_FMT[ty], v) if ty in INT_FMT else v.encode() + b"\0")
        body = struct.pack("<iiBBHHHiiii", r["ref"], r["pos"], len(name), r.get("mapq", 60), 0, len(r["cigar"]), r["flag"], len(seq), -1, -1, 0) + name + cig + bytes(packed) + bytes(r["qual"]) + tags
        raw += struct.pack("<i", len(body)) + body
    with open(path, "wb") as f:
        for i in range(0, len(raw), block):                      # several small blocks: records straddle block boundaries
            f.write(bgzf_block(raw[i:i + block]))
        f.write(bgzf_block(b""))                                  # the end-of-file marker block


def expected_batch(records, refs, intervals, long_mode):
    """The reference's rules (see the module docstring) applied to the record list, in plain Python."""
    iv_of = {}
    for i, (nm, a, b, c) in enumerate(intervals):
        iv_of.setdefault(nm, []).append((a, b, c))
    seeds = {}; examined = 0
    for r in records:
        fl = r["flag"]
        if fl & 4 or (long_mode and fl & 256) or r["ref"] < 0:
            continue
        nm = refs[r["ref"]][0]
        for a, b, c in iv_of.get(nm, []):
            examined += 1
            if not r["cigar"]:
                continue
            stop = r["pos"] + sum(l for l, o in r["cigar"] if o in "MDN=X") - 1
            if not (a <= r["pos"] <= b and a <= stop <= b):
                continue
            AS = [v for t, ty, v in r["tags"] if t == "AS"][0]
            al = dict(contig=c, pos=r["pos"] - a, AS=AS, rev=bool(fl & 16), primary=not fl & 256, cigar=[(l << 4) | OPS.index(o) for l, o in r["cigar"]],
                      seq=r["seq"], qual=[q + 33 for q in r["qual"]])
            which = 0 if long_mode else (0 if fl & 64 else 1)
            seeds.setdefault(r["name"], [[], []])[which].append(al)
    units = []
    for name in sorted(seeds):                                    # std::map order = byte-wise string order
        p = seeds[name]
        complete = any(a["primary"] for a in p[0]) and (long_mode or any(a["primary"] for a in p[1]))
        if complete:
            units.append((name, p))
    return units, examined, len(seeds), len(seeds) - len(units)


def check(b, names, cnt, units, examined, n_seeds, n_inc, long_mode):
    assert cnt == dict(examined=examined, seeds=n_seeds, incomplete=n_inc)
    assert names == [u[0] for u in units] and b["n_pairs"] == len(units)
    nm = 1 if long_mode else 2
    r = 0
    for name, p in units:
        for m in range(nm):
            c0, c1 = b["chain_off"][r], b["chain_off"][r + 1]
            als = p[m]
            assert c1 - c0 == len(als)
            got_as = b["chain_as"][c0:c1].tolist()
            assert got_as == sorted((a["AS"] for a in als), reverse=True)                 # AS-descending (ties: the library's order, compared as multisets below)
            key = lambda a: (a["AS"], a["contig"], a["pos"], a["rev"], tuple(a["cigar"]))
            got = sorted((int(b["chain_as"][c]), int(b["chain_contig"][c]), int(b["chain_pos"][c]), bool(b["chain_reverse"][c]),
                          tuple(b["cigar"][b["cigar_off"][c]:b["cigar_off"][c + 1]].tolist())) for c in range(c0, c1))
            assert got == sorted(key(a) for a in als)
            assert (b["chain_offset"][c0:c1] == 0).all()
            prim = b["read_primary"][r]
            assert c0 <= prim < c1
            # the read = bases / qualities of the first primary alignment after sorting
            ro0, ro1 = b["read_off"][r], b["read_off"][r + 1]
            cands = [a for a in als if a["primary"] and a["AS"] == b["chain_as"][prim]]
            assert any(bytes(b["read_bases"][ro0:ro1]).decode() == a["seq"] and b["read_quals"][ro0:ro1].tolist() == a["qual"] for a in cands)
            r += 1


def make_records(rng, n_names=60, lengths=(150,)):
    refs = [("chr6", 50000), ("chrUn", 9000), ("HLA-A*01", 4000)]
    recs = []
    nuc = "ACGT"
    for i in range(n_names):
        name = "read%03d" % int(rng.integers(0, 1000)) + ("x" if i % 7 == 0 else "")
        for mate in (0, 1):
            if i % 11 == 0 and mate == 1:
                continue                                         # a read whose mate never shows up: incomplete
            nal = 1 + int(rng.integers(0, 4))
            for k in range(nal):
                ref = int(rng.choice([0, 0, 0, 1, 2]))
                L = int(rng.choice(lengths)) if len(lengths) > 1 else lengths[0]
                clipl = int(rng.integers(0, 20)); clipr = int(rng.integers(0, 20))
                cig = [(clipl, "S")] if clipl else []
                mid = L - clipl - clipr
                if rng.random() < 0.3:
                    d = int(rng.integers(1, 5)); cig += [(mid // 2, "M"), (d, "D"), (mid - mid // 2, "M")]
                elif rng.random() < 0.3:
                    ins = int(rng.integers(1, 5)); cig += [(mid // 2, "M"), (ins, "I"), (mid - mid // 2 - ins, "=")]
                else:
                    cig += [(mid, "M")]
                if clipr:
                    cig += [(clipr, "H" if k and rng.random() < 0.3 else "S")]
                seq = "".join(nuc[j] for j in rng.integers(0, 4, sum(l for l, o in cig if o in "MIS=X")))
                pos = int(rng.integers(900, 3200)) if ref else int(rng.integers(9500, 21000))
                flag = 1 | (64 if mate == 0 else 128) | (16 if rng.random() < 0.5 else 0) | (256 if k else 0)
                if i % 13 == 0 and k == 0 and mate == 0:
                    flag |= 4                                    # unmapped primary: the read has no primary -> incomplete
                if k and rng.random() < 0.1:
                    flag |= 2048                                 # supplementary and secondary
                ty = str(rng.choice(["C", "c", "S", "i", "I"]))
                AS = int(rng.integers(20, 120))
                tags = [("NM", "C", 3), ("MD", "Z", "100A49"), ("AS", ty, AS), ("XS", "i", 17)]
                recs.append(dict(name=name, flag=flag, ref=ref, pos=pos, cigar=cig if not (i % 17 == 0 and k == 1) else [], seq=seq, qual=[int(q) for q in rng.integers(2, 41, len(seq))], tags=tags))
    rng.shuffle(recs)
    return refs, recs


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_bam_seed_extraction(pkg, tmp_path, seed):
    rng = np.random.default_rng(seed)
    refs, recs = make_records(rng)
    p = tmp_path / "t.bam"; write_bam(p, refs, recs)
    lib = C.CDLL(pkg.LIB_PATH)
    intervals = [("chr6", 10000, 20000, 0), ("HLA-A*01", 0, 3999, 1), ("chr6", 19000, 30000, 2)]      # overlapping intervals: a record can be taken twice
    for long_mode in (False, True):
        b, names, cnt = pkg.bam_extract_seeds(lib, p, intervals, long_read_mode=long_mode)
        units, examined, n_seeds, n_inc = expected_batch(recs, refs, intervals, long_mode)
        check(b, names, cnt, units, examined, n_seeds, n_inc, long_mode)
        assert len(units) > 5 and (long_mode or n_inc > 0)              # long reads: secondaries are dropped up front, every kept read is complete


def test_bam_errors(pkg, tmp_path):
    lib = C.CDLL(pkg.LIB_PATH)
    with pytest.raises(pkg.HlalaError):
        pkg.bam_extract_seeds(lib, tmp_path / "missing.bam", [])
    bad = tmp_path / "bad.bam"; bad.write_bytes(b"this is not a bam file at all, not even gzip")
    with pytest.raises(pkg.HlalaError):
        pkg.bam_extract_seeds(lib, bad, [])
    refs = [("chr6", 1000)]
    noas = [dict(name="r", flag=1 | 64, ref=0, pos=10, cigar=[(50, "M")], seq="A" * 50, qual=[30] * 50, tags=[("NM", "C", 0)])]
    p = tmp_path / "noas.bam"; write_bam(p, refs, noas)
    with pytest.raises(pkg.HlalaError, match="AS"):
        pkg.bam_extract_seeds(lib, p, [("chr6", 0, 999, 0)])
    unp = [dict(name="r", flag=0, ref=0, pos=10, cigar=[(50, "M")], seq="A" * 50, qual=[30] * 50, tags=[("AS", "C", 40)])]
    p2 = tmp_path / "unp.bam"; write_bam(p2, refs, unp)
    with pytest.raises(pkg.HlalaError, match="IsPaired"):
        pkg.bam_extract_seeds(lib, p2, [("chr6", 0, 999, 0)])
    b, names, cnt = pkg.bam_extract_seeds(lib, p2, [("chr6", 0, 999, 0)], long_read_mode=True)       # fine as a long read
    assert names == ["r"] and b["n_chains"] == 1 and b["chain_pos"].tolist() == [10]


# ---- round trip through the synthetic seed batch: batch -> BAM -> hlala_bam_extract_seeds -> batch ----
def batch_records(b, rng):
    recs = []
    for r in range(2 * b["n_pairs"]):
        ro0, ro1 = b["read_off"][r], b["read_off"][r + 1]
        seq = bytes(b["read_bases"][ro0:ro1]).decode(); qual = (b["read_quals"][ro0:ro1].astype(int) - 33).tolist()
        for c in range(b["chain_off"][r], b["chain_off"][r + 1]):
            cig = [(int(x) >> 4, OPS[int(x) & 15]) for x in b["cigar"][b["cigar_off"][c]:b["cigar_off"][c + 1]]]
            flag = 1 | (64 if r % 2 == 0 else 128) | (16 if b["chain_reverse"][c] else 0) | (0 if c == b["read_primary"][r] else 256)
            recs.append(dict(name="pair%07d" % (r // 2), flag=flag, ref=int(b["chain_contig"][c]), pos=int(b["chain_pos"][c] + b["chain_offset"][c]), cigar=cig,
                             seq=seq, qual=qual, tags=[("AS", "i", int(b["chain_as"][c]))]))
    order = rng.permutation(len(recs))
    return [recs[i] for i in order]


def canon(b, n_reads):
    out = []
    for r in range(n_reads):
        ro0, ro1 = b["read_off"][r], b["read_off"][r + 1]
        chains = sorted((-int(b["chain_as"][c]), int(b["chain_contig"][c]), int(b["chain_pos"][c] + b["chain_offset"][c]), int(b["chain_reverse"][c]), c == b["read_primary"][r],
                         tuple(b["cigar"][b["cigar_off"][c]:b["cigar_off"][c + 1]].tolist())) for c in range(b["chain_off"][r], b["chain_off"][r + 1]))
        out.append((bytes(b["read_bases"][ro0:ro1]), bytes(b["read_quals"][ro0:ro1]), chains))
    return out


def world_bam(tmp_path, seed, n_pairs):
    from tools import synth
    w = synth.make_world(seed=seed, G=4000, k=1)
    b = synth.make_batch(w, n_pairs, seed=seed + 5)
    clen = np.diff(w["contigs"]["contig_off"]) if len(w["contigs"]["contig_off"]) == w["contigs"]["n_contigs"] + 1 else None
    nct = w["contigs"]["n_contigs"]
    lens = [int(clen[i]) if clen is not None else 10 ** 6 for i in range(nct)]
    refs = [("hap%d" % i, lens[i]) for i in range(nct)]
    p = tmp_path / "world.bam"
    write_bam(p, refs, batch_records(b, np.random.default_rng(seed)), block=20000)
    intervals = [("hap%d" % i, 0, lens[i] - 1, i) for i in range(nct)]
    return w, b, p, intervals


def test_bam_round_trip_of_synthetic_batch(pkg, tmp_path):
    w, b, p, intervals = world_bam(tmp_path, 4, 120)
    got, names, cnt = pkg.bam_extract_seeds(C.CDLL(pkg.LIB_PATH), p, intervals)
    assert names == ["pair%07d" % i for i in range(b["n_pairs"])] and cnt["incomplete"] == 0 and cnt["examined"] == b["n_chains"]
    assert got["n_chains"] == b["n_chains"]
    assert canon(got, 2 * b["n_pairs"]) == canon(b, 2 * b["n_pairs"])
    for r in range(2 * b["n_pairs"]):                                        # AS-descending inside every read
        a = got["chain_as"][got["chain_off"][r]:got["chain_off"][r + 1]]
        assert (np.diff(a) <= 0).all()


def test_names_that_share_a_hash_are_told_apart(pkg, tmp_path, monkeypatch):
    """The decoder groups the records of a read by a 64-bit hash of its name and looks at the names once per record; different names with one hash (forced here:
    HLALA_BAM_TEST_HASH_BITS=0 keeps the partition byte only: 400 names on 256 hashes) are ordered by name within their run: the same sample as with full hashes."""
    rng = np.random.default_rng(77)
    refs, recs = make_records(rng, n_names=400)
    p = tmp_path / "t.bam"; write_bam(p, refs, recs)
    lib = C.CDLL(pkg.LIB_PATH)
    intervals = [("chr6", 10000, 20000, 0), ("HLA-A*01", 0, 3999, 1), ("chr6", 19000, 30000, 2)]
    for long_mode in (False, True):
        monkeypatch.delenv("HLALA_BAM_TEST_HASH_BITS", raising=False)
        want = pkg.bam_extract_seeds(lib, p, intervals, long_read_mode=long_mode, threads=3)
        monkeypatch.setenv("HLALA_BAM_TEST_HASH_BITS", "0")
        got = pkg.bam_extract_seeds(lib, p, intervals, long_read_mode=long_mode, threads=3)
        assert got[1] == want[1] and got[2] == want[2] and len(want[1]) > 100
        assert sorted(got[0]) == sorted(want[0])
        for k in want[0]:
            assert np.array_equal(np.asarray(got[0][k]), np.asarray(want[0][k])), k


def test_inflate_engines_decode_the_same_sample(pkg, tmp_path):
    """The BGZF blocks go through libdeflate where the machine has it and through zlib otherwise (HLALA_BAM_ZLIB=1: always): the same sample either way.
    The other engine runs in a process of its own (the choice is made once per process)."""
    import hashlib, subprocess, sys, textwrap
    w, b, p, intervals = world_bam(tmp_path, 5, 300)
    prog = textwrap.dedent("""
        import sys, ctypes as C, hashlib, importlib, json
        sys.path.insert(0, %r)
        pkg = importlib.import_module("hla-la_amd")
        lib = C.CDLL(pkg.LIB_PATH)
        lib.hlala_bam_inflate_engine.restype = C.c_char_p
        got, names, cnt = pkg.bam_extract_seeds(lib, %r, %r, threads=3)
        h = hashlib.sha256()
        for k in sorted(got):
            v = got[k]
            h.update(k.encode()); h.update(v.tobytes() if hasattr(v, "tobytes") else repr(v).encode())
        h.update(repr(names).encode()); h.update(repr(sorted(cnt.items())).encode())
        print(json.dumps([lib.hlala_bam_inflate_engine().decode(), h.hexdigest()]))
    """) % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), str(p), [tuple(i) for i in intervals])
    import json
    out = {}
    for label, env in (("default", {}), ("zlib", {"HLALA_BAM_ZLIB": "1"})):
        e = dict(os.environ); e.pop("HLALA_BAM_ZLIB", None); e.update(env)
        r = subprocess.run([sys.executable, "-c", prog], env=e, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        out[label] = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["zlib"][0] == "zlib" and out["default"][0] in ("libdeflate", "zlib")
    assert out["zlib"][1] == out["default"][1]
    lib = C.CDLL(pkg.LIB_PATH); lib.hlala_bam_inflate_engine.restype = C.c_char_p
    assert lib.hlala_bam_inflate_engine().decode() in ("libdeflate", "zlib")


def test_packed_bases_of_the_decoder_and_the_packer(pkg, tmp_path):
    """HLALA_SEEDS_PACKED: the sample's bases stay 4-bit packed as the BAM records hold them, every read on a byte of its own at (base offset + read number + 1) >> 1
    (include/hlala_gpu.h: hlala_batch_in::read_bases_packed).  Unpacked, the sample equals the one decoded to ASCII -- as a whole and window by window (odd read
    lengths, windows that start on odd reads) --, and hlala_pack_bases writes the same bytes from the ASCII bases."""
    rng = np.random.default_rng(11)
    refs, recs = make_records(rng, lengths=(149, 150, 151, 97))
    p = tmp_path / "t.bam"; write_bam(p, refs, recs)
    lib = C.CDLL(pkg.LIB_PATH)
    intervals = [("chr6", 10000, 20000, 0), ("HLA-A*01", 0, 3999, 1), ("chr6", 19000, 30000, 2)]
    for long_mode in (False, True):
        A = pkg.bam_open_seeds(lib, p, intervals, long_read_mode=long_mode)
        P = pkg.bam_open_seeds(lib, p, intervals, long_read_mode=long_mode, flags=pkg.SEEDS_PACKED)
        assert A.n_units == P.n_units and A.n_units > 5
        a = A.to_dict(); q = P.to_dict()
        assert len(set(np.diff(a["read_off"]) % 2)) == 2                     # odd and even read lengths
        for k in a:
            assert np.array_equal(np.asarray(a[k]), np.asarray(q[k])), k
        for u0, n in ((1, 3), (2, A.n_units - 2), (A.n_units - 1, 1)):
            wa = A.to_dict(u0, n); wq = P.to_dict(u0, n)
            assert P.window(u0, n).read_bases_packed and not P.window(u0, n).read_bases and P.window(u0, n).first_read == u0 * (1 if long_mode else 2)
            for k in wa:
                assert np.array_equal(np.asarray(wa[k]), np.asarray(wq[k])), (u0, n, k)
        # the host packer: same layout
        ro = a["read_off"].astype(np.int64); nr = len(ro) - 1
        out = np.zeros((int(ro[-1]) + nr + 1) // 2 + 2, np.uint8)
        lib.hlala_pack_bases.argtypes = [pkg.c_u8p, pkg.c_i64p, C.c_int64, pkg.c_u8p]
        assert lib.hlala_pack_bases(np.ascontiguousarray(a["read_bases"]).ctypes.data_as(pkg.c_u8p), ro.ctypes.data_as(pkg.c_i64p), nr, out.ctypes.data_as(pkg.c_u8p)) == 0
        d = P.window(0, P.n_units)
        ref = np.ctypeslib.as_array(d.read_bases_packed, (len(out),))
        for r in range(nr):
            at = (int(ro[r]) + r + 1) >> 1; nb = (int(ro[r + 1] - ro[r]) + 1) // 2; full = int(ro[r + 1] - ro[r]) // 2
            assert np.array_equal(out[at:at + full], ref[at:at + full]) and (nb == full or (out[at + full] >> 4) == (ref[at + full] >> 4)), r
        A.close(); P.close()


@pytest.mark.gpu
def test_packed_batch_aligns_like_the_ascii_batch(pkg, oracle, tmp_path):
    """A batch handed over with 4-bit packed bases (unpacked on the device) gives the results of the ASCII batch: through the decoder (HLALA_SEEDS_PACKED, whole
    sample and a window that starts on an odd unit) and through hlala_pack_bases on a synthetic batch."""
    from test_gpu_align import assert_pairs_equal
    w, b, p, intervals = world_bam(tmp_path, 6, 250)
    lib = C.CDLL(pkg.LIB_PATH)
    A = pkg.bam_open_seeds(lib, p, intervals); P = pkg.bam_open_seeds(lib, p, intervals, flags=pkg.SEEDS_PACKED)
    ctx = pkg.Context(w["graph"], w["contigs"], insert_mean=b["insert_mean"], insert_sd=b["insert_sd"], rng_seed=5)
    for u0, n in ((0, A.n_units), (7, 101)):
        ga = ctx.batch_window(A, u0, n); ga.align(); gp = ctx.batch_window(P, u0, n); gp.align()
        assert_pairs_equal(gp.pairs(), ga.pairs())
        ca, cp = ga.chains(1), gp.chains(1)
        for k in ca:
            if isinstance(ca[k], np.ndarray):
                assert np.array_equal(ca[k], cp[k]), k
        assert gp.stats().n_errors == 0
        ga.close(); gp.close()
    A.close(); P.close()


@pytest.mark.gpu
def test_bam_to_alignment_matches_oracle(pkg, oracle, tmp_path):
    """BAM file -> seed batch (host) -> the full alignment path on the GPU, against the oracle on the same extracted batch."""
    from test_gpu_align import assert_pairs_equal, run_both
    from util import compare_chains
    w, b, p, intervals = world_bam(tmp_path, 6, 250)
    got, names, cnt = pkg.bam_extract_seeds(C.CDLL(pkg.LIB_PATH), p, intervals)
    assert got["n_pairs"] == b["n_pairs"]
    got["insert_mean"], got["insert_sd"] = b["insert_mean"], b["insert_sd"]
    exp, gb, ctx = run_both(pkg, oracle, w, got)
    compare_chains(gb.chains(1), exp["ext"], got["n_chains"], label="bam stage B")
    assert_pairs_equal(gb.pairs(), exp["pairs"])
    assert gb.stats().n_errors == 0 and (exp["pairs"]["pair_status"] == 0).sum() > 100
