"""CPU: host logic of libhlala_gpu.so (the one-time graph flatten) against the oracle, and the C-ABI symbol table."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from tools import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def hostlib(pkg):
    so = os.environ.get("HLALA_HOST_LIB") or os.path.join(ROOT, "hla-la_amd", "libhlala_host.so")       # (tools/asan_host.sh points it at the sanitizer build)
    if not os.path.exists(so):
        import subprocess
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "hla-la_amd", "csrc"), "../libhlala_host.so"])
    lib = C.CDLL(so)
    lib.hlala_host_flatten.restype = C.c_void_p
    lib.hlala_host_flatten.argtypes = [C.POINTER(pkg.GraphDesc), C.POINTER(pkg.ContigsDesc)]
    lib.hlala_host_free.argtypes = [C.c_void_p]
    lib.hlala_host_info.argtypes = [C.c_void_p, C.POINTER(pkg.GraphInfo)]
    lib.hlala_host_paths.argtypes = [C.c_void_p, pkg.c_i32p, pkg.c_i32p, pkg.c_i32p]
    lib.hlala_host_gap_stretch.argtypes = [C.c_void_p, pkg.c_u8p]
    lib.hlala_host_jumps.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, pkg.c_i32p, pkg.c_i32p]
    lib.hlala_host_last_error.restype = C.c_char_p
    return lib


@pytest.mark.parametrize("seed,G,k", [(1, 5000, 1), (2, 8000, 0), (3, 8000, 3), (4, 3000, 10), (5, 20000, 2), (5, 2400, -1)],
                         ids=["k1", "k0", "k3", "k10", "k2", "fan-320-edges-150-jumps"])
def test_flatten_matches_oracle(pkg, oracle, hostlib, seed, G, k):
    # k = -1: nodes with 320 out- / in-edges and 150 gap-path jumps in either direction (allele-rich levels of a real PRG; no degree limit)
    w = synth.make_world(seed=seed, G=G, k=k) if k >= 0 else synth.make_fan_world(seed=seed, G=G)
    g, k1 = pkg.fill_struct(pkg.GraphDesc, w["graph"]); c, k2 = pkg.fill_struct(pkg.ContigsDesc, w["contigs"])
    F = hostlib.hlala_host_flatten(C.byref(g), C.byref(c))
    assert F, hostlib.hlala_host_last_error()
    o = oracle(w["graph"], w["contigs"])
    gi = pkg.GraphInfo(); hostlib.hlala_host_info(F, C.byref(gi)); oi = o.graph_info()
    for f, _ in pkg.GraphInfo._fields_:
        assert getattr(gi, f) == getattr(oi, f), f
    # completedGapEdgePaths in the same order, gap-stretch bitmap identical
    a = [np.zeros(gi.n_paths, np.int32) for _ in range(3)]
    hostlib.hlala_host_paths(F, *[x.ctypes.data_as(pkg.c_i32p) for x in a])
    for x, y in zip(a, o.graph_paths()):
        assert np.array_equal(x, y)
    gs = np.zeros(gi.n_levels - 1, np.uint8); hostlib.hlala_host_gap_stretch(F, gs.ctypes.data_as(pkg.c_u8p))
    assert np.array_equal(gs, o.graph_gap_stretch())
    # jump tables: per first node, targets ascending in creation index (std::map<Node*,Edge*> order)
    first, last, _ = a
    if k < 0:
        assert gi.max_out_degree == 320 and gi.max_in_degree == 320 and gi.max_jumps == 150 and gi.max_parallel == 1
    for node in np.unique(first)[:200]:
        t = np.zeros(256, np.int32); p = np.zeros(256, np.int32)
        n = hostlib.hlala_host_jumps(F, int(node), 1, 256, t.ctypes.data_as(pkg.c_i32p), p.ctypes.data_as(pkg.c_i32p))
        exp = sorted(last[first == node].tolist())
        assert n == len(exp) and t[:min(n, 256)].tolist() == exp[:256]
        assert all(first[pp] == node for pp in p[:min(n, 256)])
    hostlib.hlala_host_free(F)


@pytest.mark.parametrize("kind", ["simple", "k0-identical", "graph_m"])
def test_linear_steps_and_run_lengths(pkg, hostlib, kind):
    """FlatGraph::lin_label / lin_eid / lin_out / lin_in (what decides which DP calls the band kernel takes) against a definition written down independently
    in numpy: a step l -> l + 1 is linear when both levels hold one node, one to four parallel edges join them, no label is '_' and no gap-path jump leaves or
    enters along it; lin_label packs the labels in creation order, one per byte; the run lengths count consecutive linear steps ahead / behind, capped at 255."""
    if kind == "simple":
        w = synth.make_world(seed=7, G=6000, k=2)
    elif kind == "k0-identical":
        w = synth.make_world(seed=8, G=3000, k=0, n_mut=0, n_largegap=0)
    else:
        w = synth.make_world_m(seed=4, n_levels=60000, n_windows=2, alleles=(50, 200))
    gd = w["graph"]
    g, k1 = pkg.fill_struct(pkg.GraphDesc, gd); c, k2 = pkg.fill_struct(pkg.ContigsDesc, w["contigs"])
    F = hostlib.hlala_host_flatten(C.byref(g), C.byref(c))
    assert F, hostlib.hlala_host_last_error()
    L = gd["n_levels"]
    lab = np.zeros(L, np.uint32); eid = np.zeros(L, np.int32); lo = np.zeros(L, np.uint8); li = np.zeros(L, np.uint8)
    c_u32p = C.POINTER(C.c_uint32)
    hostlib.hlala_host_linear.argtypes = [C.c_void_p, c_u32p, pkg.c_i32p, pkg.c_u8p, pkg.c_u8p]
    hostlib.hlala_host_linear(F, lab.ctypes.data_as(c_u32p), eid.ctypes.data_as(pkg.c_i32p), lo.ctypes.data_as(pkg.c_u8p), li.ctypes.data_as(pkg.c_u8p))
    nl = gd["node_level"]; ef = gd["edge_from"]; el = gd["edge_label"]
    npl = np.bincount(nl, minlength=L); elv = nl[ef]; epl = np.bincount(elv, minlength=L)
    gapl = np.bincount(elv, weights=(el == ord('_')), minlength=L) > 0
    # levels with a '_' edge start or carry gap paths; non-gap edges out of a single node have none (checked against the jump tables below)
    one = np.zeros(L, bool); one[:-1] = (npl[:-1] == 1) & (npl[1:] == 1) & (epl[:-1] >= 1) & (epl[:-1] <= 4)
    one &= ~gapl
    exp_lab = np.zeros(L, np.uint32); exp_eid = np.full(L, -1, np.int32)
    order = np.argsort(elv, kind="stable")                      # edges by level, creation order inside a level
    start = np.concatenate([[0], np.cumsum(epl)])
    for k in range(4):
        has = one & (epl > k)
        exp_lab[has] |= el[order[start[:-1][has] + k]].astype(np.uint32) << (8 * k)
    exp_eid[one] = order[start[:-1][one]]
    assert np.array_equal(lab, exp_lab) and np.array_equal(eid, exp_eid)
    if kind == "graph_m":
        assert (one & (epl > 1)).sum() > 50          # (the SNPs of the merged backbone are parallel edges between single nodes)
    # no gap-path jump at either end of a linear step
    gi = pkg.GraphInfo(); hostlib.hlala_host_info(F, C.byref(gi))
    if gi.n_paths:
        a = [np.zeros(gi.n_paths, np.int32) for _ in range(3)]
        hostlib.hlala_host_paths(F, *[x.ctypes.data_as(pkg.c_i32p) for x in a])
        assert not one[nl[a[0]]].any()                       # first node of a path: a level whose step ahead is linear has no forward jump
        lastl = nl[a[1]]; assert not one[lastl[lastl > 0] - 1].any()
    exp_out = np.zeros(L, np.int64); exp_in = np.zeros(L, np.int64); run = 0
    for x in range(L - 1, -1, -1):
        run = min(255, run + 1) if one[x] else 0; exp_out[x] = run
    run = 0
    for x in range(L):
        run = min(255, run + 1) if (x > 0 and one[x - 1]) else 0; exp_in[x] = run
    assert np.array_equal(lo, exp_out) and np.array_equal(li, exp_in)
    if kind != "graph_m":
        assert one.mean() > 0.5          # (the stand-in graphs are mostly linear between their variants)
    hostlib.hlala_host_free(F)


def test_flatten_rejects_bad_graphs(pkg, hostlib):
    w = synth.make_world(seed=1, G=100, k=1)
    bad = dict(w["graph"]); bad["edge_to"] = bad["edge_to"].copy(); bad["edge_to"][0] = bad["edge_from"][0]      # same level
    g, _ = pkg.fill_struct(pkg.GraphDesc, bad)
    assert not hostlib.hlala_host_flatten(C.byref(g), None)
    assert b"consecutive" in hostlib.hlala_host_last_error()


def test_abi_exports_every_declared_symbol(pkg):
    hdr = open(os.path.join(ROOT, "include", "hlala_gpu.h")).read()
    declared = set(re.findall(r"\b(hlala_[a-z_0-9]+)\s*\(", hdr))
    lib = C.CDLL(pkg.LIB_PATH)          # loads without a GPU: no compute call is made
    for sym in sorted(declared):
        assert hasattr(lib, sym), f"{sym} declared in include/hlala_gpu.h but not exported"
    assert declared == set(pkg.EXPORTED_SYMBOLS)


def test_ctypes_mirror_matches_the_compiled_struct_layout(pkg):
    lib = C.CDLL(pkg.LIB_PATH)
    lib.hlala_abi_sizeof.argtypes = [C.c_char_p]; lib.hlala_abi_sizeof.restype = C.c_int
    mirror = {"hlala_graph_desc": pkg.GraphDesc, "hlala_contigs_desc": pkg.ContigsDesc, "hlala_params": pkg.Params,
              "hlala_graph_info": pkg.GraphInfo, "hlala_batch_in": pkg.BatchIn, "hlala_seeds_in": pkg.SeedsIn,
              "hlala_chains_out": pkg.ChainsOut, "hlala_pairs_out": pkg.PairsOut, "hlala_batch_stats": pkg.BatchStats,
              "hlala_exon_in": pkg.ExonIn, "hlala_call_out": pkg.CallOut, "hlala_locus_desc": pkg.LocusDesc,
              "hlala_exon_positions_out": pkg.ExonPositionsOut, "hlala_filter_params": pkg.FilterParams, "hlala_filter_stats": pkg.FilterStats, "hlala_insert_size_out": pkg.InsertSizeOut,
              "hlala_locus_info": pkg.LocusInfo, "hlala_locus_report_in": pkg.LocusReportIn, "hlala_locus_report_out": pkg.LocusReportOut, "hlala_unit_stats_out": pkg.UnitStatsOut, "hlala_pairs_packed_out": pkg.PairsPackedOut}
    for name, cls in mirror.items():
        assert lib.hlala_abi_sizeof(name.encode()) == C.sizeof(cls), name
    assert lib.hlala_abi_sizeof(b"no_such_struct") == -1


def test_create_fails_loudly_without_gpu(pkg):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    w = synth.make_world(seed=1, G=200, k=1)
    with pytest.raises(pkg.HlalaError) as e:
        pkg.Context(w["graph"], w["contigs"])
    assert "no CPU fallback" in str(e.value) or "no HIP device" in str(e.value)
