"""Damaged input files: the host-side parsers (BAM / BGZF, graph.txt, graph cache, contig files, segment / exon files) must answer with
an error or a valid result, never crash or read out of bounds.  The same test runs under AddressSanitizer with
HLALA_LIB_PATH=<a sanitizer build of the host sources> (tools/asan_host.sh)."""
import ctypes as C
import os

import numpy as np
import pytest

from test_bam import make_records, write_bam
from test_graph_files import write_contigs_dir, write_graph_txt
from tools import synth


@pytest.fixture(scope="module")
def plib(pkg):
    return C.CDLL(pkg.LIB_PATH)


def mutations(data, rng, n=30):
    yield data[:0]
    for cut in sorted(set(int(x) for x in rng.integers(1, len(data), n // 2))):
        yield data[:cut]
    for _ in range(n):
        b = bytearray(data)
        for pos in rng.integers(0, len(b), int(rng.integers(1, 6))):
            b[pos] = int(rng.integers(0, 256))
        yield bytes(b)


def test_damaged_bam_files(pkg, plib, tmp_path):
    rng = np.random.default_rng(4)
    refs, recs = make_records(rng, n_names=25)
    good = tmp_path / "g.bam"; write_bam(good, refs, recs, block=900)
    data = good.read_bytes()
    intervals = [("chr6", 10000, 20000, 0), ("HLA-A*01", 0, 3999, 1)]
    ok = err = 0
    for i, m in enumerate(mutations(data, rng)):
        p = tmp_path / "m.bam"; p.write_bytes(m)
        for long_mode in (False, True):
            try:
                b, names, cnt = pkg.bam_extract_seeds(plib, p, intervals, long_read_mode=long_mode)
                assert b["n_chains"] == len(b["chain_pos"]) and len(names) == b["n_pairs"]
                ok += 1
            except pkg.HlalaError:
                err += 1
    assert err > 10          # truncations and broken blocks are noticed (CRC / sizes); harmless flips may pass


def test_bgzf_block_size_smaller_than_header(pkg, plib, tmp_path):
    """A BC field below the size of the block's own header (here BSIZE = 5) used to underflow the payload length and abort the process
    with an exception thrown across the C boundary; huge header / record lengths must be refused the same way."""
    import struct
    rng = np.random.default_rng(5)
    refs, recs = make_records(rng, n_names=5)
    good = tmp_path / "g.bam"; write_bam(good, refs, recs, block=900)
    data = bytearray(good.read_bytes())
    assert data[12:14] == b"BC"
    intervals = [("chr6", 10000, 20000, 0), ("HLA-A*01", 0, 3999, 1)]
    for bsize in (0, 5, 17, 24):
        m = bytearray(data); m[16:18] = struct.pack("<H", bsize)
        p = tmp_path / "b.bam"; p.write_bytes(bytes(m))
        with pytest.raises(pkg.HlalaError):
            pkg.bam_extract_seeds(plib, p, intervals)
    # l_text of 2 GB in an otherwise valid first block
    from test_bam import bgzf_block
    raw = b"BAM\x01" + struct.pack("<i", 0x7FFFFFF0) + b"x" * 64
    p = tmp_path / "t.bam"; p.write_bytes(bgzf_block(raw) + bgzf_block(b""))
    with pytest.raises(pkg.HlalaError):
        pkg.bam_extract_seeds(plib, p, intervals)


def test_damaged_graph_and_contig_files(pkg, plib, tmp_path):
    rng = np.random.default_rng(5)
    w = synth.make_world(seed=2, G=120, k=1)
    gtxt = tmp_path / "graph.txt"; write_graph_txt(gtxt, w["graph"], rng, pipe_level=3)
    cache = tmp_path / "graph.cache"; pkg.save_graph_cache(plib, pkg.load_graph_text(plib, gtxt), cache)
    for src, loader in ((gtxt, pkg.load_graph_text), (cache, pkg.load_graph_cache)):
        data = src.read_bytes(); bad = 0
        for m in mutations(data, rng, 40):
            p = tmp_path / "m.bin"; p.write_bytes(m)
            try:
                g = loader(plib, p)
                assert len(g["edge_from"]) == g["n_edges"] == len(g["edge_label"])
            except pkg.HlalaError:
                bad += 1
        assert bad > 5
    d = tmp_path / "dir"; d.mkdir(); write_contigs_dir(d, w["contigs"], rng)
    for name in ("sequences.txt", "translation/1.txt", "mapping_PRGonly/referenceGenome.fa"):
        orig = (d / name).read_bytes()
        for m in mutations(orig, rng, 16):
            (d / name).write_bytes(m)
            try:
                c, iv = pkg.load_contigs_dir(plib, d, False)
                assert len(c["contig_seq"]) == len(c["contig_level"]) == c["contig_off"][-1]
            except pkg.HlalaError:
                pass
        (d / name).write_bytes(orig)


def test_damaged_segment_and_exon_files(pkg, plib, tmp_path):
    from test_typer_files import make_graph_dir
    rng = np.random.default_rng(6)
    make_graph_dir(tmp_path, rng, n_types=10)
    prg = tmp_path / "PRG"
    for name in ("segments.txt", "3_gene_HLA-A_2_exon_2.txt", "5_gene_HLA-A_4_exon_3.txt"):
        orig = (prg / name).read_bytes()
        for m in mutations(orig, rng, 16):
            (prg / name).write_bytes(m)
            try:
                T = pkg.Typer(plib, tmp_path)
                try:
                    L = T.locus("A"); assert L.cluster_seq.shape == (L.n_clusters, L.n_columns); L.free()
                except pkg.HlalaError:
                    pass
                T.close()
            except pkg.HlalaError:
                pass
        (prg / name).write_bytes(orig)
