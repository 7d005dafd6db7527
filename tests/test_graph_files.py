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
