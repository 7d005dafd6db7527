"""PRG/graph.txt loader and the binary graph cache (Graph/Graph.cpp:2225-2545, LocusCodeAllocation.cpp:264-312)."""
import ctypes as C
import os

import numpy as np
import pytest

from tools import synth


def write_graph_txt(path, g, rng, pipe_level=None):
    """A graph.txt as Graph::writeToFile produces it (Graph.cpp:2225-2327): per-locus allele codes (here: one locus per level, alleles
    coded 1.. in alphabetical order as LocusCodeAllocation::doCode assigns them), arbitrary 1-based node / edge indices, label and
    pgf_protect fields.  Nodes and edges are written in creation order, so the loaded arrays must equal the source arrays.
    `pipe_level`: use the byte '|' as the code of one allele at that level -- the "|||||||" case of the reader (:2338-2365)."""
    lev_of_edge = g["node_level"][g["edge_from"]]
    codes = {}
    for lvl in np.unique(lev_of_edge):
        alleles = sorted(set(bytes(g["edge_label"][lev_of_edge == lvl]).decode()))
        codes[int(lvl)] = {a: (ord("|") if (lvl == pipe_level and i == 0) else 1 + i) for i, a in enumerate(alleles)}
    node_idx = rng.permutation(g["n_nodes"]) + 1               # arbitrary distinct indices: only the order of the lines matters
    with open(path, "wb") as f:
        f.write(b"CODE:\n")
        for lvl in sorted(codes):
            for a, c in sorted(codes[lvl].items()):
                f.write(f"L{lvl}|||{a}|||{c}\n".encode())
        f.write(b"NODES:\n")
        for i in range(g["n_nodes"]):
            f.write(f"{node_idx[i]}|||{g['node_level'][i]}|||{int(g['node_level'][i] == g['n_levels'] - 1)}\n".encode())
        f.write(b"EDGES:\n")
        for e in range(g["n_edges"]):
            lvl = int(lev_of_edge[e]); code = codes[lvl][chr(g["edge_label"][e])]
            rec = f"{e + 1}|||L{lvl}|||1|||".encode() + bytes([code]) + f"|||{node_idx[g['edge_from'][e]]}|||{node_idx[g['edge_to'][e]]}".encode()
            if e % 3:                                           # both the 8-field and the 6-field form
                rec += b"|||" + (b"pgf" if e % 2 else b"") + b"|||" + (b"1" if e % 5 == 0 else b"0")
            f.write(rec + (b"\n" if e + 1 < g["n_edges"] else b""))


@pytest.mark.parametrize("seed,G,k", [(1, 300, 1), (2, 500, 3), (3, 400, 0)])
def test_graph_text_and_cache_round_trip(pkg, tmp_path, seed, G, k):
    g = synth.make_world(seed=seed, G=G, k=k)["graph"]
    lib = C.CDLL(pkg.LIB_PATH)
    p = tmp_path / "graph.txt"
    write_graph_txt(p, g, np.random.default_rng(seed), pipe_level=7)
    got = pkg.load_graph_text(lib, p)
    for key in ("n_levels", "n_nodes", "n_edges"):
        assert got[key] == g[key], key
    for key in ("node_level", "edge_from", "edge_to", "edge_label"):
        assert np.array_equal(got[key], g[key]), key
    c = tmp_path / "graph.cache"
    pkg.save_graph_cache(lib, got, c)
    back = pkg.load_graph_cache(lib, c)
    for key in ("node_level", "edge_from", "edge_to", "edge_label"):
        assert np.array_equal(back[key], g[key]), key
    assert os.path.getsize(c) == 20 + 4 * g["n_nodes"] + 9 * g["n_edges"]


def test_graph_text_errors(pkg, tmp_path):
    lib = C.CDLL(pkg.LIB_PATH)
    cases = {"unknown_node": "CODE:\nL0|||A|||1\nNODES:\n1|||0|||0\n2|||1|||1\nEDGES:\n1|||L0|||1|||\x01|||1|||3\n",
             "unknown_locus": "CODE:\nL0|||A|||1\nNODES:\n1|||0|||0\n2|||1|||1\nEDGES:\n1|||L9|||1|||\x01|||1|||2\n",
             "unknown_allele": "CODE:\nL0|||A|||1\nNODES:\n1|||0|||0\n2|||1|||1\nEDGES:\n1|||L0|||1|||\x02|||1|||2\n",
             "bad_fields": "CODE:\nL0|||A|||1\nNODES:\n1|||0\n", "no_header": "1|||0|||0\n", "bad_code": "CODE:\nL0|||A|||999\n"}
    for name, text in cases.items():
        p = tmp_path / (name + ".txt"); p.write_bytes(text.encode("latin-1"))
        with pytest.raises(pkg.HlalaError):
            pkg.load_graph_text(lib, p)
    with pytest.raises(pkg.HlalaError):
        pkg.load_graph_text(lib, tmp_path / "missing.txt")
    bad = tmp_path / "bad.cache"; bad.write_bytes(b"not a cache")
    with pytest.raises(pkg.HlalaError):
        pkg.load_graph_cache(lib, bad)
    ok = "CODE:\nL0|||A|||1\nL0|||C|||2\nNODES:\n7|||0|||0\n3|||1|||1\nEDGES:\n1|||L0|||1|||\x01|||7|||3\n2|||L0|||1|||\x02|||7|||3|||lab|||0"
    p = tmp_path / "ok.txt"; p.write_bytes(ok.encode("latin-1"))
    g = pkg.load_graph_text(lib, p)
    assert g["n_levels"] == 2 and g["node_level"].tolist() == [0, 1] and g["edge_from"].tolist() == [0, 0] and bytes(g["edge_label"]) == b"AC"


def write_contigs_dir(root, contigs, rng, trailing_newline=True):
    """sequences.txt / FASTA files / translation files of a graph directory for the contigs of a synthetic world."""
    n = contigs["n_contigs"]; off = contigs["contig_off"]
    (root / "translation").mkdir(); (root / "mapping_PRGonly").mkdir(); (root / "extendedReferenceGenome").mkdir()
    rows = ["SequenceID\tName\tFASTAID\tChr\tStart_1based\tStop_1based"]
    ext = [">chrUn some description\n" + "ACGTN" * 40 + "\n"]; prg = []
    chr6 = "".join("ACGT"[x] for x in rng.integers(0, 4, 777)); starts = {}
    for i in range(n):
        sid = int(contigs["contig_seqid"][i]); s = bytes(contigs["contig_seq"][off[i]:off[i + 1]]).decode()
        if i == 0:
            starts[i] = len(chr6) + 1; chr6 += s + "GATTACA" * 11
            rows.append("%d\tref\tpgf\tchr6\t%d\t%d" % (sid, starts[i], starts[i] + len(s) - 1))
            prg.append(">chr6\n" + s + "\n")                            # PRG-only mode: the whole sequence is the interval
        else:
            rows.append("%d\talt%d\talt%d\t\t\t" % (sid, i, i))
            wrapped = "\n".join(s[j:j + 61] for j in range(0, len(s), 61))
            ext.append(">PRG_%d extra words\n%s\n" % (sid, wrapped)); prg.append(">PRG_%d\n%s\n" % (sid, wrapped))
        lv = contigs["contig_level"][off[i]:off[i + 1]]
        (root / "translation" / ("%d.txt" % sid)).write_text("\n".join(str(int(x)) for x in lv) + ("\n" if trailing_newline or i % 2 else ""))
    ext.insert(1, ">chr6\n" + "\n".join(chr6[j:j + 70] for j in range(0, len(chr6), 70)) + "\n")
    (root / "sequences.txt").write_text("\n".join(rows) + "\n")
    (root / "extendedReferenceGenome" / "extendedReferenceGenome.fa").write_text("".join(ext))
    (root / "mapping_PRGonly" / "referenceGenome.fa").write_text("".join(prg))
    return starts


@pytest.mark.parametrize("extended,trailing", [(True, True), (False, True), (True, False)])
def test_contigs_directory_loader(pkg, tmp_path, extended, trailing):
    """sequences.txt + FASTA + translation files -> hlala_contigs_desc (processBAM.cpp:1183-1402, :4389-4457), including the extra level-0
    entry a translation file ending in a newline produces in the reference."""
    w = synth.make_world(seed=3, G=1500, k=1)
    c = w["contigs"]; n = c["n_contigs"]; off = c["contig_off"]
    rng = np.random.default_rng(1)
    starts = write_contigs_dir(tmp_path, c, rng, trailing_newline=trailing)
    lib = C.CDLL(pkg.LIB_PATH)
    got, intervals = pkg.load_contigs_dir(lib, tmp_path, extended)
    assert got["n_contigs"] == n and np.array_equal(got["contig_seqid"], c["contig_seqid"])
    for i in range(n):
        extra = 1 if (trailing or i % 2) else 0
        a, b = got["contig_off"][i], got["contig_off"][i + 1]
        assert b - a == off[i + 1] - off[i] + extra
        assert np.array_equal(got["contig_seq"][a:b - extra if extra else b], c["contig_seq"][off[i]:off[i + 1]])
        assert np.array_equal(got["contig_level"][a:b - extra if extra else b], c["contig_level"][off[i]:off[i + 1]])
        if extra:
            assert got["contig_seq"][b - 1] == ord("N") and got["contig_level"][b - 1] == 0
        L = int(off[i + 1] - off[i])
        if i == 0:
            assert intervals[i] == (("chr6", starts[0] - 1, starts[0] + L - 2, 0) if extended else ("chr6", 0, L - 1, 0))
        else:
            assert intervals[i] == ("PRG_%d" % c["contig_seqid"][i], 0, L - 1, i)
    # the same in two steps (what the host program does: the BAM decoder starts on the intervals while the translation tables are still being read)
    h = C.c_void_p()
    lib.hlala_contigs_open_dir.argtypes = [C.c_char_p, C.c_int32, C.POINTER(C.c_void_p)]
    lib.hlala_contigs_load_translations.argtypes = [C.c_void_p]; lib.hlala_contigs_file_free.argtypes = [C.c_void_p]; lib.hlala_contigs_file_free.restype = None
    lib.hlala_contigs_file_desc.argtypes = [C.c_void_p, C.POINTER(pkg.ContigsDesc)]; lib.hlala_contigs_file_intervals.argtypes = [C.c_void_p, C.POINTER(pkg.BamInterval), C.c_int32]
    assert lib.hlala_contigs_open_dir(str(tmp_path).encode(), int(extended), C.byref(h)) == 0
    iv = (pkg.BamInterval * n)()
    assert lib.hlala_contigs_file_intervals(h, iv, n) == n
    assert [(iv[i].ref_name.decode(), iv[i].start_0based, iv[i].stop_0based, iv[i].contig) for i in range(n)] == intervals
    d = pkg.ContigsDesc()
    assert lib.hlala_contigs_file_desc(h, C.byref(d)) != 0                 # not before the translation tables are in
    assert lib.hlala_contigs_load_translations(h) == 0 and lib.hlala_contigs_load_translations(h) == 0
    assert lib.hlala_contigs_file_desc(h, C.byref(d)) == 0 and d.n_contigs == n
    tot = int(got["contig_off"][-1])
    assert np.array_equal(np.ctypeslib.as_array(d.contig_off, (n + 1,)), got["contig_off"])
    assert np.array_equal(np.ctypeslib.as_array(d.contig_level, (tot,)), got["contig_level"]) and np.array_equal(np.ctypeslib.as_array(d.contig_seq, (tot,)), got["contig_seq"])
    lib.hlala_contigs_file_free(h)
    # errors
    (tmp_path / "translation" / ("%d.txt" % c["contig_seqid"][1])).unlink()
    with pytest.raises(pkg.HlalaError, match="translation"):
        pkg.load_contigs_dir(lib, tmp_path, extended)
    h = C.c_void_p()
    assert lib.hlala_contigs_open_dir(str(tmp_path).encode(), int(extended), C.byref(h)) == 0       # (the intervals do not need the tables)
    assert lib.hlala_contigs_load_translations(h) != 0
    lib.hlala_contigs_file_free(h)
    with pytest.raises(pkg.HlalaError):
        pkg.load_contigs_dir(lib, tmp_path / "nowhere", extended)
