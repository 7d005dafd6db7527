"""ctypes binding of libhlala_gpu.so (the C ABI in include/hlala_gpu.h).

This package is plumbing for tests and bench.py: the product is the shared library built from
csrc/ (hand-written HIP for gfx950 behind a plain C ABI).  There is NO CPU fallback here: if
the library or a GPU is missing, loading/creating fails loudly.

The directory name has a hyphen (repo layout contract), so import it through
`importlib` -- see `tests/conftest.py` / `__graft_entry__.py` (`load_package()`).
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# HLALA_LIB_PATH: another build of the same library (tools/asan_host.sh points it at a sanitizer build of the host-side sources)
LIB_PATH = os.environ.get("HLALA_LIB_PATH") or os.path.join(_HERE, "libhlala_gpu.so")

c_i32p = C.POINTER(C.c_int32)
c_i64p = C.POINTER(C.c_int64)
c_u8p = C.POINTER(C.c_uint8)
c_u32p = C.POINTER(C.c_uint32)
c_f64p = C.POINTER(C.c_double)

CHAIN_OK, CHAIN_SKIP_STRAND, CHAIN_SKIP_DUP = 0, 1, 2
CHAIN_ERR_COLUMNS, CHAIN_ERR_FRONTIER, CHAIN_ERR_INPUT = -1, -2, -3


class GraphDesc(C.Structure):
    _fields_ = [("n_levels", C.c_int32), ("n_nodes", C.c_int32), ("n_edges", C.c_int32),
                ("node_level", c_i32p), ("edge_from", c_i32p), ("edge_to", c_i32p), ("edge_label", c_u8p)]


class ContigsDesc(C.Structure):
    _fields_ = [("n_contigs", C.c_int32), ("contig_off", c_i64p), ("contig_seq", c_u8p),
                ("contig_level", c_i32p), ("contig_seqid", c_i32p)]


class Params(C.Structure):
    _fields_ = [("insert_mean", C.c_double), ("insert_sd", C.c_double), ("rng_seed", C.c_uint32),
                ("long_read_mode", C.c_int32), ("max_columns", C.c_int32), ("reserved", C.c_int32)]


class GraphInfo(C.Structure):
    _fields_ = [("n_levels", C.c_int32), ("n_nodes", C.c_int32), ("n_edges", C.c_int32), ("n_paths", C.c_int32),
                ("n_jump_entries", C.c_int64), ("n_path_edges", C.c_int64), ("n_levelpos_entries", C.c_int64),
                ("max_nodes_per_level", C.c_int32), ("max_out_degree", C.c_int32), ("max_in_degree", C.c_int32),
                ("n_gap_stretch_levels", C.c_int32), ("max_jumps", C.c_int32), ("max_parallel", C.c_int32)]


class BatchIn(C.Structure):
    # (64-bit offsets: a batch is a window into the arrays of a sample, include/hlala_gpu.h)
    _fields_ = [("n_pairs", C.c_int32), ("read_off", c_i64p), ("read_bases", c_u8p), ("read_quals", c_u8p),
                ("chain_off", c_i64p), ("read_primary", c_i32p), ("n_chains", C.c_int32),
                ("chain_contig", c_i32p), ("chain_pos", c_i32p), ("chain_offset", c_i32p), ("chain_as", c_i32p),
                ("chain_reverse", c_u8p), ("cigar_off", c_i64p), ("cigar", c_u32p), ("read_bases_packed", c_u8p), ("first_read", C.c_int64)]


class SeedsIn(C.Structure):
    _fields_ = [("n_reads", C.c_int32), ("read_off", c_i32p), ("read_bases", c_u8p), ("read_quals", c_u8p),
                ("n_chains", C.c_int32), ("chain_read", c_i32p), ("chain_seq_begin", c_i32p),
                ("chain_seq_end", c_i32p), ("chain_reverse", c_u8p), ("col_off", c_i32p), ("col_level", c_i32p),
                ("col_edge", c_i32p), ("col_gchar", c_u8p), ("col_schar", c_u8p)]


class ChainsOut(C.Structure):
    _fields_ = [("status", c_i32p), ("n_cols", c_i32p), ("seq_begin", c_i32p), ("seq_end", c_i32p),
                ("removed_cols", c_i32p), ("ll", c_f64p), ("dp_iters", c_i32p), ("dp_score", c_i32p),
                ("col_level", c_i32p), ("col_edge", c_i32p), ("col_gchar", c_u8p), ("col_schar", c_u8p),
                ("col_fromseed", c_u8p)]


class PairsOut(C.Structure):
    _fields_ = [("pair_status", c_i32p), ("best_chain", c_i32p), ("n_combinations", c_i32p),
                ("pair_ll", c_f64p), ("pair_mapq", c_f64p), ("mate_mapq", c_f64p), ("strands_valid", c_u8p),
                ("n_cols", c_i32p), ("col_level", c_i32p), ("col_edge", c_i32p), ("col_gchar", c_u8p),
                ("col_schar", c_u8p), ("col_fromseed", c_u8p), ("col_mapq", c_u8p)]


class ExonIn(C.Structure):
    _fields_ = [("n_clusters", C.c_int32), ("exon_length", C.c_int32), ("cluster_seq", c_u8p), ("n_reads", C.c_int32),
                ("pos_off", c_i32p), ("pos_exon", c_i32p), ("pos_g0", c_u8p), ("pos_glen", c_i32p), ("pos_qual", c_u8p),
                ("pos_use", c_u8p)]


E_CAPACITY = -4      # HLALA_E_CAPACITY


class LocusDesc(C.Structure):
    _fields_ = [("level_min", C.c_int32), ("level_max", C.c_int32), ("level_to_exon", c_i32p), ("insert_mean", C.c_double), ("insert_sd", C.c_double),
                ("min_mapq", C.c_double), ("min_weighted_ok", C.c_double), ("pair_mask", c_u8p), ("min_alignment_columns", C.c_int32), ("reserved", C.c_int32)]


class ExonPositionsOut(C.Structure):
    _fields_ = [("cap_reads", C.c_int32), ("cap_pos", C.c_int32), ("cap_chars", C.c_int32), ("n_reads", C.c_int32), ("n_pos", C.c_int32), ("n_chars", C.c_int32),
                ("n_pairs_ok", C.c_int32), ("n_pairs_broken", C.c_int32),
                ("read_pair", c_i32p), ("read_weighted_ok", c_f64p), ("read_fraction_ok", c_f64p), ("read_distance", c_i32p), ("read_cols_nongap", c_i32p),
                ("pos_off", c_i32p), ("pos_exon", c_i32p), ("pos_level", c_i32p), ("pos_mate", c_u8p), ("pos_mapq", c_u8p), ("pos_novel_gap", c_i32p),
                ("geno_off", c_i32p), ("geno_chars", c_u8p), ("qual_chars", c_u8p), ("read_reverse", c_u8p), ("read_mapq", c_f64p)]


def alloc_exon_positions_out(cap_reads, cap_pos, cap_chars):
    """ExonPositionsOut over fresh numpy arrays; returns (struct, dict of arrays)."""
    d = dict(read_pair=np.zeros(cap_reads, np.int32), read_weighted_ok=np.zeros(2 * cap_reads), read_fraction_ok=np.zeros(2 * cap_reads),
             read_distance=np.zeros(cap_reads, np.int32), read_cols_nongap=np.zeros(2 * cap_reads, np.int32), pos_off=np.zeros(cap_reads + 1, np.int32),
             pos_exon=np.zeros(cap_pos, np.int32), pos_level=np.zeros(cap_pos, np.int32), pos_mate=np.zeros(cap_pos, np.uint8), pos_mapq=np.zeros(cap_pos, np.uint8),
             pos_novel_gap=np.zeros(cap_pos, np.int32), geno_off=np.zeros(cap_pos + 1, np.int32), geno_chars=np.zeros(cap_chars, np.uint8), qual_chars=np.zeros(cap_chars, np.uint8),
             read_reverse=np.zeros(2 * cap_reads, np.uint8), read_mapq=np.zeros(2 * cap_reads))
    o = ExonPositionsOut(); o.cap_reads, o.cap_pos, o.cap_chars = cap_reads, cap_pos, cap_chars
    for k, v in d.items():
        setattr(o, k, v.ctypes.data_as(dict(ExonPositionsOut._fields_)[k]))
    return o, d


def trim_exon_positions(o, d):
    """Cut the arrays of alloc_exon_positions_out to what the call filled in."""
    nr, npos, nch = o.n_reads, o.n_pos, o.n_chars
    cut = dict(read_pair=nr, read_weighted_ok=2 * nr, read_fraction_ok=2 * nr, read_distance=nr, read_cols_nongap=2 * nr, pos_off=nr + 1, pos_exon=npos, pos_level=npos,
               pos_mate=npos, pos_mapq=npos, pos_novel_gap=npos, geno_off=npos + 1, geno_chars=nch, qual_chars=nch, read_reverse=2 * nr, read_mapq=2 * nr)
    out = {k: d[k][:n].copy() for k, n in cut.items()}
    out.update(n_reads=nr, n_pos=npos, n_chars=nch, n_pairs_ok=o.n_pairs_ok, n_pairs_broken=o.n_pairs_broken)
    return out


def make_locus_desc(level_min, level_to_exon, insert_mean, insert_sd, min_mapq=0.0, min_weighted_ok=0.0, pair_mask=None, min_alignment_columns=1000):
    l2e = np.ascontiguousarray(level_to_exon, np.int32)
    L = LocusDesc(); L.level_min = int(level_min); L.level_max = int(level_min) + len(l2e) - 1; L.level_to_exon = l2e.ctypes.data_as(c_i32p)
    L.insert_mean, L.insert_sd, L.min_mapq, L.min_weighted_ok = float(insert_mean), float(insert_sd), float(min_mapq), float(min_weighted_ok)
    L.min_alignment_columns = int(min_alignment_columns)
    keep = [l2e]
    if pair_mask is not None:
        m = np.ascontiguousarray(pair_mask, np.uint8); L.pair_mask = m.ctypes.data_as(c_u8p); keep.append(m)
    return L, keep


class FilterParams(C.Structure):
    _fields_ = [("filter_first20", C.c_int32), ("first20_n", C.c_int32), ("first20_min_prop", C.c_double), ("first20_limit_per_read", C.c_int32),
                ("min_per_position_mapq", C.c_double), ("high_coverage_filter", C.c_int32), ("high_coverage_min_coverage", C.c_int32), ("high_coverage_min_freq", C.c_double),
                ("long_read_strand_filter", C.c_int32), ("strand_min_allele_coverage", C.c_int32), ("strand_min_freq", C.c_double)]


class FilterStats(C.Structure):
    _fields_ = [(k, C.c_int64) for k in ("considered_positions", "positions_with_removed_alleles", "considered_alleles", "removed_alleles", "reads_kicked_out",
                                         "reads_kicked_out_robust", "high_coverage_positions", "high_coverage_removed_alleles", "bases_used",
                                         "strand_alleles_enough_coverage", "strand_removed_alleles", "strand_positions_with_removed")]


def default_filter_params(**kw):
    """The reference's settings for short reads (hla/HLATyper.cpp:28-31, 69-75; HLATyper.h:57)."""
    p = FilterParams(1, 20, 0.1, 2, 0.7, 0, 100, 0.2, 0, 100, 0.1)          # long reads: long_read_strand_filter=1
    for k, v in kw.items():
        setattr(p, k, v)
    return p


def exon_positions_struct(e):
    """ExonPositionsOut view over the dict Batch.exon_positions() returns (for hlala_filter_positions)."""
    o = ExonPositionsOut(); keep = []
    o.n_reads, o.n_pos, o.n_chars = int(e["n_reads"]), int(e["n_pos"]), int(e["n_chars"])
    o.cap_reads, o.cap_pos, o.cap_chars = o.n_reads, o.n_pos, o.n_chars
    types = dict(ExonPositionsOut._fields_)
    for k in ("read_pair", "read_weighted_ok", "read_fraction_ok", "read_distance", "read_cols_nongap", "pos_off", "pos_exon", "pos_level", "pos_mate", "pos_mapq",
              "pos_novel_gap", "geno_off", "geno_chars", "qual_chars", "read_reverse", "read_mapq"):
        if k not in e:
            continue                                       # optional arrays stay NULL
        a = np.ascontiguousarray(e[k]); keep.append(a); setattr(o, k, a.ctypes.data_as(types[k]))
    return o, keep


def filter_positions(lib, e, params=None):
    """hlala_filter_positions on the dict Batch.exon_positions() returns: (pos_use, read_ignored, stats dict)."""
    o, keep = exon_positions_struct(e)
    prm = params or default_filter_params()
    use = np.zeros(max(1, o.n_pos), np.uint8); ign = np.zeros(max(1, o.n_reads), np.uint8); st = FilterStats()
    lib.hlala_filter_positions.argtypes = [C.POINTER(ExonPositionsOut), C.POINTER(FilterParams), c_u8p, c_u8p, C.POINTER(FilterStats)]
    rc = lib.hlala_filter_positions(C.byref(o), C.byref(prm), use.ctypes.data_as(c_u8p), ign.ctypes.data_as(c_u8p), C.byref(st))
    if rc != 0:
        raise HlalaError(f"hlala_filter_positions failed ({rc})")
    return use[:o.n_pos], ign[:o.n_reads], {k: int(getattr(st, k)) for k, _ in FilterStats._fields_}


def _graph_from_handle(lib, h):
    d = GraphDesc()
    lib.hlala_graph_file_desc.argtypes = [C.c_void_p, C.POINTER(GraphDesc)]
    if lib.hlala_graph_file_desc(h, C.byref(d)) != 0:
        raise HlalaError("hlala_graph_file_desc failed")
    g = dict(n_levels=d.n_levels, n_nodes=d.n_nodes, n_edges=d.n_edges,
             node_level=np.ctypeslib.as_array(d.node_level, (d.n_nodes,)).copy(), edge_from=np.ctypeslib.as_array(d.edge_from, (max(d.n_edges, 1),))[:d.n_edges].copy(),
             edge_to=np.ctypeslib.as_array(d.edge_to, (max(d.n_edges, 1),))[:d.n_edges].copy(), edge_label=np.ctypeslib.as_array(d.edge_label, (max(d.n_edges, 1),))[:d.n_edges].copy())
    lib.hlala_graph_file_free.argtypes = [C.c_void_p]; lib.hlala_graph_file_free.restype = None
    lib.hlala_graph_file_free(h)
    return g


def load_graph_text(lib, path):
    """PRG/graph.txt -> graph dict (hlala_graph_load_text)."""
    h = C.c_void_p(); lib.hlala_graph_load_text.argtypes = [C.c_char_p, C.POINTER(C.c_void_p)]; lib.hlala_loader_last_error.restype = C.c_char_p
    if lib.hlala_graph_load_text(str(path).encode(), C.byref(h)) != 0:
        raise HlalaError(lib.hlala_loader_last_error().decode(errors="replace"))
    return _graph_from_handle(lib, h)


def save_graph_cache(lib, graph, path):
    s, keep = fill_struct(GraphDesc, graph)
    lib.hlala_graph_cache_save.argtypes = [C.POINTER(GraphDesc), C.c_char_p]; lib.hlala_loader_last_error.restype = C.c_char_p
    if lib.hlala_graph_cache_save(C.byref(s), str(path).encode()) != 0:
        raise HlalaError(lib.hlala_loader_last_error().decode(errors="replace"))


def load_graph_cache(lib, path):
    h = C.c_void_p(); lib.hlala_graph_cache_load.argtypes = [C.c_char_p, C.POINTER(C.c_void_p)]; lib.hlala_loader_last_error.restype = C.c_char_p
    if lib.hlala_graph_cache_load(str(path).encode(), C.byref(h)) != 0:
        raise HlalaError(lib.hlala_loader_last_error().decode(errors="replace"))
    return _graph_from_handle(lib, h)


def load_contigs_dir(lib, graph_dir, extended_reference_genome=True):
    """hlala_contigs_load_dir: (contigs dict for Context, intervals [(ref name, start0, stop0, contig)] for bam_extract_seeds)."""
    h = C.c_void_p()
    lib.hlala_contigs_load_dir.argtypes = [C.c_char_p, C.c_int32, C.POINTER(C.c_void_p)]; lib.hlala_loader_last_error.restype = C.c_char_p
    if lib.hlala_contigs_load_dir(str(graph_dir).encode(), int(bool(extended_reference_genome)), C.byref(h)) != 0:
        raise HlalaError(lib.hlala_loader_last_error().decode(errors="replace"))
    d = ContigsDesc()
    lib.hlala_contigs_file_desc.argtypes = [C.c_void_p, C.POINTER(ContigsDesc)]; lib.hlala_contigs_file_desc(h, C.byref(d))
    n = d.n_contigs
    off = np.ctypeslib.as_array(d.contig_off, (n + 1,)).astype(np.int64).copy(); tot = int(off[-1])
    contigs = dict(n_contigs=n, contig_off=off, contig_seq=np.ctypeslib.as_array(d.contig_seq, (max(tot, 1),))[:tot].astype(np.uint8).copy(),
                   contig_level=np.ctypeslib.as_array(d.contig_level, (max(tot, 1),))[:tot].astype(np.int32).copy(),
                   contig_seqid=np.ctypeslib.as_array(d.contig_seqid, (max(n, 1),))[:n].astype(np.int32).copy())
    iv = (BamInterval * max(1, n))()
    lib.hlala_contigs_file_intervals.argtypes = [C.c_void_p, C.POINTER(BamInterval), C.c_int32]
    lib.hlala_contigs_file_intervals(h, iv, n)
    intervals = [(iv[i].ref_name.decode(), iv[i].start_0based, iv[i].stop_0based, iv[i].contig) for i in range(n)]
    lib.hlala_contigs_file_free.argtypes = [C.c_void_p]; lib.hlala_contigs_file_free.restype = None
    lib.hlala_contigs_file_free(h)
    return contigs, intervals


class BamInterval(C.Structure):
    _fields_ = [("ref_name", C.c_char_p), ("start_0based", C.c_int32), ("stop_0based", C.c_int32), ("contig", C.c_int32)]


class SeedBatch:
    """hlala_seed_batch handle: the seeds of a whole sample (64-bit offsets) as decoded from a BAM file."""

    def __init__(self, lib, h, long_read_mode):
        self.lib, self.h, self.long_read_mode = lib, h, bool(long_read_mode)
        lib.hlala_seed_batch_units.argtypes = [C.c_void_p]; lib.hlala_seed_batch_units.restype = C.c_int64
        lib.hlala_seed_batch_window.argtypes = [C.c_void_p, C.c_int64, C.c_int32, C.POINTER(BatchIn)]
        lib.hlala_seed_batch_name.argtypes = [C.c_void_p, C.c_int64]; lib.hlala_seed_batch_name.restype = C.c_char_p
        lib.hlala_seed_batch_free.argtypes = [C.c_void_p]; lib.hlala_seed_batch_free.restype = None
        lib.hlala_seed_batch_timing.argtypes = [C.c_void_p, c_f64p, c_i32p]
        lib.hlala_seed_batch_desc.argtypes = [C.c_void_p, C.POINTER(BatchIn), C.POINTER(C.c_int64)]
        self.n_units = int(lib.hlala_seed_batch_units(h))
        cnt = (C.c_int64 * 3)()
        lib.hlala_seed_batch_counts.argtypes = [C.c_void_p, C.POINTER(C.c_int64)]
        lib.hlala_seed_batch_counts(h, cnt)          # (hlala_seed_batch_desc would fill in the whole sample: the windows are filled as they are handed out)
        self.counts = dict(examined=int(cnt[0]), seeds=int(cnt[1]), incomplete=int(cnt[2]))

    def window(self, first_unit, n_units) -> "BatchIn":
        """units [first_unit, first_unit + n_units) as a batch descriptor pointing into the handle (no copy; valid while the handle lives)"""
        d = BatchIn()
        if self.lib.hlala_seed_batch_window(self.h, int(first_unit), int(n_units), C.byref(d)) != 0:
            raise HlalaError(self.lib.hlala_bam_last_error().decode(errors="replace"))
        return d

    def to_dict(self, first_unit=0, n_units=None):
        """a window copied into numpy arrays with offsets starting at 0 (the layout the tests and the oracle binding use); first_chain = its first absolute chain"""
        n = self.n_units - first_unit if n_units is None else n_units
        d = self.window(first_unit, n)
        nr = n * (1 if self.long_read_mode else 2); nc = d.n_chains

        def arr(ptr, off, cnt, dt):
            if cnt == 0:
                return np.zeros(0, dt)
            return np.ctypeslib.as_array(ptr, (off + cnt,))[off:off + cnt].astype(dt).copy()
        ro = arr(d.read_off, 0, nr + 1, np.int64); co = arr(d.chain_off, 0, nr + 1, np.int64)
        r0, c0 = int(ro[0]), int(co[0])
        go = arr(d.cigar_off, c0, nc + 1, np.int64); g0 = int(go[0])
        if d.read_bases_packed:        # a sample decoded with HLALA_SEEDS_PACKED: unpacked here for the tests (read R of the sample starts at byte (base offset + R + 1) >> 1)
            R0 = int(d.first_read); p0 = (r0 + R0 + 1) >> 1; p1 = ((int(ro[-1]) + R0 + nr + 1) >> 1) + 1
            pk = arr(d.read_bases_packed, p0, p1 - p0, np.uint8); lut = np.frombuffer(b"=ACMGRSVTWYHKDBN", np.uint8)
            bases = np.zeros(int(ro[-1]) - r0, np.uint8)
            for r in range(nr):
                a, z = int(ro[r]) - r0, int(ro[r + 1]) - r0; at = ((int(ro[r]) + R0 + r + 1) >> 1) - p0
                by = pk[at:at + (z - a + 1) // 2]; nib = np.empty(2 * len(by), np.uint8); nib[0::2] = by >> 4; nib[1::2] = by & 15
                bases[a:z] = lut[nib[:z - a]]
        else:
            bases = arr(d.read_bases, r0, int(ro[-1]) - r0, np.uint8)
        return dict(n_pairs=n, read_off=(ro - r0), read_bases=bases, read_quals=arr(d.read_quals, r0, int(ro[-1]) - r0, np.uint8),
                    chain_off=(co - c0), read_primary=arr(d.read_primary, 0, nr, np.int32) - c0, n_chains=nc,
                    chain_contig=arr(d.chain_contig, c0, nc, np.int32), chain_pos=arr(d.chain_pos, c0, nc, np.int32), chain_offset=arr(d.chain_offset, c0, nc, np.int32),
                    chain_as=arr(d.chain_as, c0, nc, np.int32), chain_reverse=arr(d.chain_reverse, c0, nc, np.uint8), cigar_off=(go - g0),
                    cigar=arr(d.cigar, g0, int(go[-1]) - g0, np.uint32), first_chain=c0)

    def names(self, first_unit=0, n_units=None):
        n = self.n_units - first_unit if n_units is None else n_units
        return [self.lib.hlala_seed_batch_name(self.h, first_unit + i).decode() for i in range(n)]

    def timing(self):
        s = np.zeros(6, np.float64); t = np.zeros(1, np.int32)
        self.lib.hlala_seed_batch_timing(self.h, s.ctypes.data_as(c_f64p), t.ctypes.data_as(c_i32p))
        return dict(zip(("index", "inflate", "parse", "group", "name_sort", "layout"), [float(x) for x in s]), threads=int(t[0]))

    def pin(self, on=True):
        self.lib.hlala_seed_batch_pin.argtypes = [C.c_void_p, C.c_int]
        return self.lib.hlala_seed_batch_pin(self.h, 1 if on else 0) == 0

    def close(self):
        if self.h:
            self.lib.hlala_seed_batch_free(self.h); self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


SEEDS_PACKED = 1          # HLALA_SEEDS_PACKED


def bam_open_seeds(lib, path, intervals, long_read_mode=False, threads=0, flags=0) -> SeedBatch:
    """hlala_bam_extract_seeds_opt: intervals = [(ref name, start_0based, stop_0based, contig index)]; returns the handle of the whole sample."""
    arr = (BamInterval * max(1, len(intervals)))()
    for i, (nm, a, b, c) in enumerate(intervals):
        arr[i] = BamInterval(nm.encode(), int(a), int(b), int(c))
    h = C.c_void_p()
    lib.hlala_bam_extract_seeds_opt.argtypes = [C.c_char_p, C.c_int32, C.POINTER(BamInterval), C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_void_p)]
    lib.hlala_bam_last_error.restype = C.c_char_p
    if lib.hlala_bam_extract_seeds_opt(str(path).encode(), len(intervals), arr, int(bool(long_read_mode)), int(threads), int(flags), C.byref(h)) != 0:
        raise HlalaError(lib.hlala_bam_last_error().decode(errors="replace"))
    return SeedBatch(lib, h, long_read_mode)


def bam_extract_seeds(lib, path, intervals, long_read_mode=False, threads=0):
    """The whole sample as one batch dict (offsets from 0), the read names and the counters; for samples that fit one batch."""
    S = bam_open_seeds(lib, path, intervals, long_read_mode, threads)
    try:
        b = S.to_dict(); b.pop("first_chain")
        return b, S.names(), S.counts
    finally:
        S.close()


def exon_in_from_positions(e, pos_use, cluster_seq, n_clusters, exon_length):
    """hlala_exon_in (input of hlala_exon_loglik) from the outputs of hlala_exon_positions and hlala_filter_positions: the likelihood loop
    reads the first genotype character, the genotype length and the first quality of every position (hla/HLATyper.cpp:2080-2277)."""
    go = np.asarray(e["geno_off"], np.int64)
    n = int(e["n_pos"])
    first = go[:n]
    g0 = np.asarray(e["geno_chars"], np.uint8)[first] if n else np.zeros(0, np.uint8)
    q0 = np.asarray(e["qual_chars"], np.uint8)[first] if n else np.zeros(0, np.uint8)
    return dict(n_clusters=int(n_clusters), exon_length=int(exon_length), cluster_seq=np.ascontiguousarray(cluster_seq, np.uint8).reshape(-1),
                n_reads=int(e["n_reads"]), pos_off=np.ascontiguousarray(e["pos_off"], np.int32), pos_exon=np.ascontiguousarray(e["pos_exon"], np.int32),
                pos_g0=np.ascontiguousarray(g0), pos_glen=np.ascontiguousarray(go[1:n + 1] - go[:n], np.int32), pos_qual=np.ascontiguousarray(q0),
                pos_use=np.ascontiguousarray(pos_use, np.uint8))


class LocusInfo(C.Structure):
    _fields_ = [("n_clusters", C.c_int32), ("n_columns", C.c_int32), ("n_exons", C.c_int32), ("level_min", C.c_int32), ("level_max", C.c_int32), ("n_types", C.c_int32),
                ("cluster_seq", c_u8p), ("level_to_exon", c_i32p), ("col_level", c_i32p), ("col_exon", c_i32p), ("col_exon_pos", c_i32p), ("exon_length", c_i32p)]


class PairsPackedOut(C.Structure):
    _fields_ = [("cap_cols", C.c_int64), ("n_cols_total", C.c_int64), ("col_off", c_i64p), ("col_level", c_i32p), ("col_edge", c_i32p), ("col_gchar", c_u8p),
                ("col_schar", c_u8p), ("col_fromseed", c_u8p), ("col_mapq", c_u8p)]


class UnitStatsOut(C.Structure):
    _fields_ = [("valid", c_u8p), ("strands_valid", c_u8p), ("distance", c_i32p), ("fraction_ok", c_f64p), ("weighted_ok", c_f64p), ("n_columns", c_i32p), ("mate_mapq", c_f64p)]


def unit_stats_struct(d):
    """UnitStatsOut view over a dict of arrays (valid, strands_valid, distance, fraction_ok, weighted_ok, n_columns, mate_mapq)."""
    o = UnitStatsOut(); keep = []
    dt = dict(valid=np.uint8, strands_valid=np.uint8, distance=np.int32, fraction_ok=np.float64, weighted_ok=np.float64, n_columns=np.int32, mate_mapq=np.float64)
    for k, t in UnitStatsOut._fields_:
        a = np.ascontiguousarray(d[k], dt[k]); keep.append(a); setattr(o, k, a.ctypes.data_as(t))
    return o, keep


def typer_write_summary(lib, out_dir, stats, unpaired=False, unit_mask=None, insert_mean=0.0, insert_sd=0.0, min_alignment_length_unpaired=1000):
    """summaryStatistics.txt (hlala_typer_write_summary) from the dict Batch.unit_stats() returns."""
    o, keep = unit_stats_struct(stats)
    n = len(stats["valid"]); m = None if unit_mask is None else np.ascontiguousarray(unit_mask, np.uint8)
    lib.hlala_typer_write_summary.argtypes = [C.c_char_p, C.c_int32, C.c_int32, c_u8p, C.POINTER(UnitStatsOut), C.c_double, C.c_double, C.c_int32]
    lib.hlala_typer_last_error.restype = C.c_char_p
    if lib.hlala_typer_write_summary(str(out_dir).encode(), n, int(bool(unpaired)), None if m is None else m.ctypes.data_as(c_u8p), C.byref(o), insert_mean, insert_sd,
                                     min_alignment_length_unpaired) != 0:
        raise HlalaError(lib.hlala_typer_last_error().decode(errors="replace"))


class LocusReportIn(C.Structure):
    _fields_ = [("pos", C.POINTER(ExonPositionsOut)), ("filter", C.POINTER(FilterParams)), ("unit_name_1", C.POINTER(C.c_char_p)), ("unit_name_2", C.POINTER(C.c_char_p)),
                ("long_read_mode", C.c_int32), ("n_clusters", C.c_int32), ("pair_ll", c_f64p), ("mis_avg", c_f64p), ("mis_min", c_f64p), ("order", c_i32p),
                ("p_normalized", c_f64p), ("call", C.c_void_p), ("kmers_covered", C.c_double * 2), ("unaccounted_min_coverage", C.c_int32), ("pairs_file_done", C.c_int32),
                ("unaccounted_min_fraction", C.c_double), ("unit_stats", C.POINTER(UnitStatsOut)), ("unit_mask", c_u8p), ("n_units", C.c_int32), ("reserved2", C.c_int32),
                ("insert_mean", C.c_double), ("insert_sd", C.c_double), ("min_mapq", C.c_double), ("min_weighted_ok", C.c_double)]


class LocusReportOut(C.Structure):
    _fields_ = [("locus_coverage", C.c_double), ("first_decile_coverage", C.c_double), ("minimum_coverage", C.c_double), ("avg_column_error", C.c_double),
                ("min_column_p", C.c_double), ("bases_used", C.c_int64), ("n_columns_unaccounted", C.c_int32), ("n_utilized_reads", C.c_int32),
                ("n_piled_positions", C.c_int32), ("reserved", C.c_int32)]


class Typer:
    """hlala_typer wrapper (host code): graph level names, gene level ranges, exon files and allele clusters of a locus, result files."""

    def __init__(self, lib, graph_dir):
        self.lib = lib; self.h = C.c_void_p()
        lib.hlala_typer_open.argtypes = [C.c_char_p, C.POINTER(C.c_void_p)]
        lib.hlala_typer_last_error.restype = C.c_char_p
        self._check(lib.hlala_typer_open(str(graph_dir).encode(), C.byref(self.h)))
        lib.hlala_typer_n_levels.argtypes = [C.c_void_p]; lib.hlala_typer_level_name.argtypes = [C.c_void_p, C.c_int32]; lib.hlala_typer_level_name.restype = C.c_char_p
        lib.hlala_typer_level_of.argtypes = [C.c_void_p, C.c_char_p]; lib.hlala_typer_n_genes.argtypes = [C.c_void_p]
        lib.hlala_typer_gene.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_char_p), c_i32p, c_i32p]
        lib.hlala_typer_load_g_groups.argtypes = [C.c_void_p, C.c_char_p]
        lib.hlala_typer_locus.argtypes = [C.c_void_p, C.c_char_p, C.c_int32, C.POINTER(C.c_char_p), C.POINTER(C.c_void_p)]
        lib.hlala_typer_close.argtypes = [C.c_void_p]; lib.hlala_typer_close.restype = None

    def _check(self, rc):
        if rc != 0:
            raise HlalaError(self.lib.hlala_typer_last_error().decode(errors="replace"))

    def level_names(self):
        return [self.lib.hlala_typer_level_name(self.h, i).decode() for i in range(self.lib.hlala_typer_n_levels(self.h))]

    def level_of(self, name):
        return self.lib.hlala_typer_level_of(self.h, name.encode())

    def genes(self):
        """[(gene, first level, last level)] in name order: the intervals of hlala_set_gene_intervals."""
        out = []
        for i in range(self.lib.hlala_typer_n_genes(self.h)):
            nm = C.c_char_p(); a = C.c_int32(); b = C.c_int32()
            self._check(self.lib.hlala_typer_gene(self.h, i, C.byref(nm), C.byref(a), C.byref(b)))
            out.append((nm.value.decode(), a.value, b.value))
        return out

    def load_g_groups(self, path):
        self._check(self.lib.hlala_typer_load_g_groups(self.h, str(path).encode()))

    def g_translate(self, alleles):
        """translate_allele_list_to_G_allele for a list of allele names: (G group or joined list, perfectly)."""
        self.lib.hlala_typer_g_translate.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.c_int32, C.POINTER(C.c_int32)]
        buf = C.create_string_buffer(1 << 16); perf = C.c_int32(0)
        self._check(self.lib.hlala_typer_g_translate(self.h, ";".join(alleles).encode(), buf, len(buf), C.byref(perf)))
        return buf.value.decode(), bool(perf.value)

    def locus(self, name, exons=None):
        h = C.c_void_p()
        if exons is None:
            arr, n = None, 0
        else:
            arr = (C.c_char_p * len(exons))(*[e.encode() for e in exons]); n = len(exons)
        self._check(self.lib.hlala_typer_locus(self.h, name.encode(), n, arr, C.byref(h)))
        return Locus(self, h, name)

    def close(self):
        if self.h:
            self.lib.hlala_typer_close(self.h); self.h = None


class Locus:
    """hlala_locus wrapper: exon columns and allele clusters of one locus."""

    def __init__(self, typer, h, name):
        self.typer, self.lib, self.h, self.name = typer, typer.lib, h, name
        lib = self.lib
        lib.hlala_locus_get.argtypes = [C.c_void_p, C.POINTER(LocusInfo)]
        lib.hlala_locus_cluster_id.argtypes = [C.c_void_p, C.c_int32]; lib.hlala_locus_cluster_id.restype = C.c_char_p
        lib.hlala_locus_type_cluster.argtypes = [C.c_void_p, C.c_char_p]
        lib.hlala_locus_cluster_kmers.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_char_p, C.c_int32, c_i32p, c_i32p]
        lib.hlala_locus_write_files.argtypes = [C.c_void_p, C.POINTER(LocusReportIn), C.c_char_p, C.POINTER(LocusReportOut)]
        lib.hlala_locus_free.argtypes = [C.c_void_p]; lib.hlala_locus_free.restype = None
        i = LocusInfo(); typer._check(lib.hlala_locus_get(h, C.byref(i)))
        nl = i.level_max - i.level_min + 1
        cp = lambda ptr, n, dt: np.ctypeslib.as_array(ptr, (max(n, 1),))[:n].astype(dt).copy()
        self.n_clusters, self.n_columns, self.n_exons, self.level_min, self.level_max, self.n_types = i.n_clusters, i.n_columns, i.n_exons, i.level_min, i.level_max, i.n_types
        self.cluster_seq = cp(i.cluster_seq, i.n_clusters * i.n_columns, np.uint8).reshape(i.n_clusters, i.n_columns)
        self.level_to_exon = cp(i.level_to_exon, nl, np.int32); self.col_level = cp(i.col_level, i.n_columns, np.int32)
        self.col_exon = cp(i.col_exon, i.n_columns, np.int32); self.col_exon_pos = cp(i.col_exon_pos, i.n_columns, np.int32); self.exon_length = cp(i.exon_length, i.n_exons, np.int32)

    def cluster_id(self, c):
        return self.lib.hlala_locus_cluster_id(self.h, c).decode()

    def type_cluster(self, hla_type):
        return self.lib.hlala_locus_type_cluster(self.h, hla_type.encode())

    def cluster_kmers(self, cluster, k=31):
        """(query k-mers without '*', total number of k-mers) of a cluster's exon sequences."""
        nq = C.c_int32(); nt = C.c_int32()
        self.lib.hlala_locus_cluster_kmers(self.h, cluster, k, None, 0, C.byref(nq), C.byref(nt))
        buf = C.create_string_buffer(max(1, nq.value * k))
        self.typer._check(self.lib.hlala_locus_cluster_kmers(self.h, cluster, k, buf, nq.value, C.byref(nq), C.byref(nt)))
        raw = buf.raw[:nq.value * k].decode()
        return [raw[i * k:(i + 1) * k] for i in range(nq.value)], nt.value

    def write_files(self, out_dir, e, names1, names2, pair_ll, mis_avg, mis_min, order, p_normalized, call, kmers_covered=(-1.0, -1.0), params=None,
                    long_read_mode=False, unaccounted_min_coverage=30, unaccounted_min_fraction=0.2, unit_stats=None, unit_mask=None, insert_mean=0.0, insert_sd=0.0,
                    min_mapq=0.0, min_weighted_ok=0.0):
        """hlala_locus_write_files; e = dict of Batch.exon_positions(), call = CallOut, unit_stats = dict of Batch.unit_stats() (histogram lines).
        Returns LocusReportOut."""
        o, keep = exon_positions_struct(e)
        prm = params or default_filter_params()
        r = LocusReportIn(); r.pos = C.pointer(o); r.filter = C.pointer(prm)
        n1 = (C.c_char_p * max(1, len(names1)))(*[n.encode() for n in names1]); r.unit_name_1 = n1
        if names2 is not None:
            n2 = (C.c_char_p * max(1, len(names2)))(*[n.encode() for n in names2]); r.unit_name_2 = n2
        r.long_read_mode = int(bool(long_read_mode)); r.n_clusters = self.n_clusters
        arrs = [np.ascontiguousarray(pair_ll, np.float64), np.ascontiguousarray(mis_avg, np.float64), np.ascontiguousarray(mis_min, np.float64),
                np.ascontiguousarray(order, np.int32), np.ascontiguousarray(p_normalized, np.float64)]
        r.pair_ll, r.mis_avg, r.mis_min = (a.ctypes.data_as(c_f64p) for a in arrs[:3]); r.order = arrs[3].ctypes.data_as(c_i32p); r.p_normalized = arrs[4].ctypes.data_as(c_f64p)
        r.call = C.cast(C.pointer(call), C.c_void_p); r.kmers_covered[0], r.kmers_covered[1] = float(kmers_covered[0]), float(kmers_covered[1])
        r.unaccounted_min_coverage = unaccounted_min_coverage; r.unaccounted_min_fraction = unaccounted_min_fraction
        if unit_stats is not None:
            us, keep_us = unit_stats_struct(unit_stats); r.unit_stats = C.pointer(us); r.n_units = len(unit_stats["valid"])
            if unit_mask is not None:
                um = np.ascontiguousarray(unit_mask, np.uint8); r.unit_mask = um.ctypes.data_as(c_u8p)
            r.insert_mean, r.insert_sd, r.min_mapq, r.min_weighted_ok = insert_mean, insert_sd, min_mapq, min_weighted_ok
        out = LocusReportOut()
        self.typer._check(self.lib.hlala_locus_write_files(self.h, C.byref(r), str(out_dir).encode(), C.byref(out)))
        return out

    def free(self):
        if self.h:
            self.lib.hlala_locus_free(self.h); self.h = None


def typer_begin_output(lib, out_dir, unaccounted_min_fraction=0.2):
    lib.hlala_typer_begin_output.argtypes = [C.c_char_p, C.c_double]; lib.hlala_typer_last_error.restype = C.c_char_p
    if lib.hlala_typer_begin_output(str(out_dir).encode(), unaccounted_min_fraction) != 0:
        raise HlalaError(lib.hlala_typer_last_error().decode(errors="replace"))


def typer_end_output(lib, out_dir, loci, very_conservative=False):
    lib.hlala_typer_end_output.argtypes = [C.c_char_p, C.c_char_p, C.c_int32]; lib.hlala_typer_last_error.restype = C.c_char_p
    if lib.hlala_typer_end_output(str(out_dir).encode(), ",".join(loci).encode(), int(very_conservative)) != 0:
        raise HlalaError(lib.hlala_typer_last_error().decode(errors="replace"))



class InsertSizeOut(C.Structure):
    _fields_ = [("mean", C.c_double), ("sd", C.c_double), ("n_used", C.c_int32), ("n_skipped", C.c_int32), ("total_weight", C.c_double)]


class CallOut(C.Structure):
    _fields_ = [("first_cluster", C.c_int32), ("second_cluster", C.c_int32), ("first_marginal", C.c_double), ("second_p", C.c_double),
                ("ll_max", C.c_double), ("max_pair", C.c_int32), ("n_sort_ties", C.c_int32)]


class BatchStats(C.Structure):
    _fields_ = [("ms_project", C.c_float), ("ms_extend", C.c_float), ("ms_pair", C.c_float),
                ("n_chains_extended", C.c_int64), ("n_dp_calls", C.c_int64), ("n_dp_iterations", C.c_int64),
                ("n_dp_cells", C.c_int64), ("n_seed_columns", C.c_int64), ("n_out_columns", C.c_int64),
                ("n_edges_touched", C.c_int64), ("n_errors", C.c_int64), ("ms_extend_retry", C.c_float),
                ("n_chains_retried", C.c_int32), ("ms_dp_main", C.c_float), ("n_dp_retried_large", C.c_int32), ("n_dp_shared", C.c_int64),
                ("n_dp_class", C.c_int32 * 7), ("ms_dp_class", C.c_float * 7), ("ms_side", C.c_float),
                ("n_dp_lane", C.c_int32), ("ms_dp_lane", C.c_float), ("n_dp_jump_free", C.c_int32), ("ms_dp_jump_free", C.c_float),
                ("n_dp_band", C.c_int32), ("n_dp_band_failed", C.c_int32), ("n_dp_jump_free_failed", C.c_int32), ("ms_dp_band", C.c_float),
                ("n_dp_band2", C.c_int32), ("n_dp_band2_failed", C.c_int32), ("ms_dp_band2", C.c_float), ("reserved_stats", C.c_int32)]


_DT = {c_i32p: np.int32, c_i64p: np.int64, c_u8p: np.uint8, c_u32p: np.uint32, c_f64p: np.float64}


def fill_struct(cls, d):
    """Build a ctypes struct from a dict of numpy arrays / scalars.  Returns (struct, keepalive)."""
    s = cls()
    keep = []
    for name, ctype in cls._fields_:
        if name not in d or d[name] is None:
            continue
        if ctype in _DT:
            a = np.ascontiguousarray(d[name], dtype=_DT[ctype])
            keep.append(a)
            setattr(s, name, a.ctypes.data_as(ctype))
        else:
            setattr(s, name, d[name])
    return s, keep


def alloc_chains_out(n_chains, stride):
    d = dict(status=np.zeros(n_chains, np.int32), n_cols=np.zeros(n_chains, np.int32),
             seq_begin=np.zeros(n_chains, np.int32), seq_end=np.zeros(n_chains, np.int32),
             removed_cols=np.zeros(n_chains, np.int32), ll=np.zeros(n_chains, np.float64),
             dp_iters=np.zeros(2 * n_chains, np.int32), dp_score=np.zeros(2 * n_chains, np.int32),
             col_level=np.zeros(n_chains * stride, np.int32), col_edge=np.zeros(n_chains * stride, np.int32),
             col_gchar=np.zeros(n_chains * stride, np.uint8), col_schar=np.zeros(n_chains * stride, np.uint8),
             col_fromseed=np.zeros(n_chains * stride, np.uint8))
    s, keep = fill_struct(ChainsOut, d)
    d["_stride"] = stride
    return s, d


def alloc_pairs_out(n_pairs, stride):
    n2 = 2 * n_pairs
    d = dict(pair_status=np.zeros(n_pairs, np.int32), best_chain=np.zeros(n2, np.int32),
             n_combinations=np.zeros(n_pairs, np.int32), pair_ll=np.zeros(n_pairs, np.float64),
             pair_mapq=np.zeros(n_pairs, np.float64), mate_mapq=np.zeros(n2, np.float64),
             strands_valid=np.zeros(n_pairs, np.uint8), n_cols=np.zeros(n2, np.int32),
             col_level=np.zeros(n2 * stride, np.int32), col_edge=np.zeros(n2 * stride, np.int32),
             col_gchar=np.zeros(n2 * stride, np.uint8), col_schar=np.zeros(n2 * stride, np.uint8),
             col_fromseed=np.zeros(n2 * stride, np.uint8), col_mapq=np.zeros(n2 * stride, np.uint8))
    s, keep = fill_struct(PairsOut, d)
    d["_stride"] = stride
    return s, d


class HlalaError(RuntimeError):
    pass


_lib = None


def load_library(path: str | None = None):
    """Load libhlala_gpu.so.  Fails loudly when the HIP extension has not been built."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    if not os.path.exists(p):
        raise HlalaError(f"{p} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                         f"(hipcc --offload-arch=gfx950); there is no CPU fallback")
    lib = C.CDLL(p)
    vp = C.c_void_p
    lib.hlala_create.argtypes = [C.POINTER(vp), C.c_int, vp, C.POINTER(GraphDesc), C.POINTER(ContigsDesc), C.POINTER(Params)]
    lib.hlala_create.restype = C.c_int
    lib.hlala_destroy.argtypes = [vp]
    lib.hlala_destroy.restype = None
    lib.hlala_last_error.argtypes = [vp]
    lib.hlala_last_error.restype = C.c_char_p
    lib.hlala_graph_get_info.argtypes = [vp, C.POINTER(GraphInfo)]
    lib.hlala_graph_get_nodes.argtypes = [vp, c_i32p, c_i32p]
    lib.hlala_graph_get_paths.argtypes = [vp, c_i32p, c_i32p, c_i32p]
    lib.hlala_graph_get_gap_stretch.argtypes = [vp, c_u8p]
    lib.hlala_batch_create.argtypes = [vp, C.POINTER(BatchIn), C.POINTER(vp)]
    lib.hlala_batch_create_unpaired.argtypes = [vp, C.POINTER(BatchIn), C.POINTER(vp)]
    lib.hlala_batch_create_from_seeds.argtypes = [vp, C.POINTER(SeedsIn), C.POINTER(vp)]
    lib.hlala_batch_destroy.argtypes = [vp]
    lib.hlala_batch_destroy.restype = None
    for f in ("hlala_project_chains", "hlala_extend_chains", "hlala_pair_chains", "hlala_align_batch"):
        getattr(lib, f).argtypes = [vp, vp]
        getattr(lib, f).restype = C.c_int
    lib.hlala_batch_get_chains.argtypes = [vp, vp, C.c_int, C.POINTER(ChainsOut)]
    lib.hlala_batch_get_pairs.argtypes = [vp, vp, C.POINTER(PairsOut)]
    lib.hlala_batch_get_stats.argtypes = [vp, vp, C.POINTER(BatchStats)]
    lib.hlala_batch_export_pair_records.argtypes = [vp, vp, vp]
    lib.hlala_set_gene_intervals.argtypes = [vp, C.c_int32, c_i32p, c_i32p]
    lib.hlala_postprocess_pairs.argtypes = [vp, vp, c_u8p]
    lib.hlala_get_coverage.argtypes = [vp, c_i32p, C.c_int]
    lib.hlala_exon_loglik.argtypes = [vp, C.POINTER(ExonIn), c_f64p, c_i32p]
    lib.hlala_pair_loglik.argtypes = [vp, c_f64p, c_i32p, C.c_int32, C.c_int32, c_f64p, c_f64p, c_f64p]
    lib.hlala_kat_phred.argtypes = [vp, C.c_int, c_f64p, c_u8p, c_u8p, c_f64p]
    lib.hlala_kat_rand_r.argtypes = [vp, C.c_int, c_u32p, c_i32p]
    lib.hlala_abi_sizeof.argtypes = [C.c_char_p]
    lib.hlala_estimate_insert_size.argtypes = [vp, C.POINTER(BatchIn), C.POINTER(InsertSizeOut)]
    lib.hlala_exon_positions.argtypes = [vp, vp, C.POINTER(LocusDesc), C.POINTER(ExonPositionsOut)]
    lib.hlala_call_locus.argtypes = [vp, C.c_int32, c_f64p, c_f64p, c_f64p, c_i32p, c_f64p, c_f64p, C.POINTER(CallOut)]
    lib.hlala_abi_sizeof.restype = C.c_int
    # a library built from another revision of include/hlala_gpu.h may keep every struct size and still mean something else by a field
    # (round 3: 64-bit window offsets in hlala_batch_in): refuse it instead of uploading garbage
    if not hasattr(lib, "hlala_abi_version") or lib.hlala_abi_version() != ABI_VERSION:
        got = lib.hlala_abi_version() if hasattr(lib, "hlala_abi_version") else "none"
        raise HlalaError(f"{p}: interface version {got}, this binding mirrors version {ABI_VERSION} of include/hlala_gpu.h -- rebuild the library")
    if path is None:
        _lib = lib
    return lib


ABI_VERSION = 5              # HLALA_ABI_VERSION of include/hlala_gpu.h
DEBUG_WC_N, DEBUG_WC_BAND_FETCH, DEBUG_WC_BAND_WHY, DEBUG_WC_BAND_TIED = 88, 48, 62, 68      # include/hlala_gpu.h: debug section
BUILD_AGENT_RELEASE = 2      # hlala_build_flags(): the in-memory DP class releases at agent scope (make EXTRA=-DHLALA_DP_AGENT_RELEASE)


EXPORTED_SYMBOLS = [
    "hlala_debug_work_counters", "hlala_debug_dp_items", "hlala_debug_counters", "hlala_debug_buffer", "hlala_debug_memory",
    "hlala_set_tail_pool", "hlala_flush",
    "hlala_comm_create", "hlala_comm_destroy", "hlala_comm_uses_rccl", "hlala_comm_last_error", "hlala_gather_pair_records", "hlala_reduce_coverage",
    "hlala_create", "hlala_destroy", "hlala_last_error", "hlala_graph_get_info", "hlala_graph_get_nodes",
    "hlala_graph_get_paths", "hlala_graph_get_gap_stretch", "hlala_batch_create",
    "hlala_batch_create_from_seeds", "hlala_batch_create_unpaired", "hlala_batch_set_first_chain", "hlala_batch_destroy", "hlala_project_chains", "hlala_extend_chains",
    "hlala_pair_chains", "hlala_align_batch", "hlala_batch_get_chains", "hlala_batch_get_pairs",
    "hlala_batch_get_stats", "hlala_batch_get_pairs_packed", "hlala_batch_export_pair_records", "hlala_set_gene_intervals", "hlala_postprocess_pairs", "hlala_get_coverage", "hlala_exon_loglik", "hlala_pair_loglik", "hlala_kat_phred",
    "hlala_kat_rand_r", "hlala_kat_exp", "hlala_abi_sizeof", "hlala_abi_version", "hlala_build_flags", "hlala_pack_bases", "hlala_call_locus", "hlala_exon_positions", "hlala_filter_positions", "hlala_estimate_insert_size", "hlala_graph_load_text", "hlala_graph_cache_save",
    "hlala_graph_cache_load", "hlala_graph_file_desc", "hlala_graph_file_free", "hlala_loader_last_error",
    "hlala_bam_extract_seeds", "hlala_bam_extract_seeds_mt", "hlala_bam_extract_seeds_opt", "hlala_seed_batch_counts", "hlala_seed_batch_desc", "hlala_seed_batch_window", "hlala_seed_batch_units", "hlala_seed_batch_name", "hlala_seed_batch_timing",
    "hlala_seed_batch_free", "hlala_seed_batch_pin", "hlala_bam_last_error", "hlala_bam_inflate_engine", "hlala_pinned_alloc", "hlala_pinned_free", "hlala_host_register", "hlala_host_unregister", "hlala_set_insert_size",
    "hlala_contigs_load_dir", "hlala_contigs_open_dir", "hlala_contigs_load_translations", "hlala_contigs_file_desc", "hlala_contigs_file_intervals", "hlala_contigs_file_free",
    "hlala_typer_open", "hlala_typer_close", "hlala_typer_last_error", "hlala_typer_n_levels", "hlala_typer_level_name", "hlala_typer_level_of", "hlala_typer_n_genes",
    "hlala_typer_gene", "hlala_typer_load_g_groups", "hlala_typer_g_translate", "hlala_typer_locus", "hlala_locus_free", "hlala_locus_get", "hlala_locus_cluster_id", "hlala_locus_type_cluster",
    "hlala_locus_cluster_kmers", "hlala_type_locus", "hlala_kmer_presence", "hlala_kmer_keep_reads", "hlala_kmer_presence_kept", "hlala_kmer_forget_reads", "hlala_unit_alignment_stats", "hlala_typer_write_summary", "hlala_typer_begin_output", "hlala_locus_write_files", "hlala_locus_write_pairs_file", "hlala_typer_end_output",
]


class Context:
    """hlala_ctx wrapper: uploads the flattened graph once (hlala_create)."""

    def __init__(self, graph: dict, contigs: dict | None, insert_mean=200.0, insert_sd=35.0, rng_seed=12345,
                 long_read_mode=0, max_columns=384, device=0, stream=None):
        self.lib = load_library()
        self.max_columns = max_columns
        g, self._kg = fill_struct(GraphDesc, graph)
        self.params = Params(insert_mean, insert_sd, rng_seed, long_read_mode, max_columns, 0)
        h = C.c_void_p()
        if contigs is not None:
            c, self._kc = fill_struct(ContigsDesc, contigs)
            cp = C.byref(c)
        else:
            cp = None
        rc = self.lib.hlala_create(C.byref(h), device, C.c_void_p(stream or 0), C.byref(g), cp, C.byref(self.params))
        if rc != 0:
            msg = self.lib.hlala_last_error(None)
            raise HlalaError(f"hlala_create failed ({rc}): {msg.decode() if msg else ''}")
        self.h = h

    def _check(self, rc, what):
        if rc != 0:
            msg = self.lib.hlala_last_error(self.h)
            raise HlalaError(f"{what} failed ({rc}): {msg.decode() if msg else ''}")

    def graph_info(self) -> GraphInfo:
        gi = GraphInfo()
        self._check(self.lib.hlala_graph_get_info(self.h, C.byref(gi)), "hlala_graph_get_info")
        return gi

    def graph_paths(self):
        gi = self.graph_info()
        a = [np.zeros(gi.n_paths, np.int32) for _ in range(3)]
        self._check(self.lib.hlala_graph_get_paths(self.h, *[x.ctypes.data_as(c_i32p) for x in a]), "hlala_graph_get_paths")
        return a

    def graph_gap_stretch(self):
        gi = self.graph_info()
        a = np.zeros(gi.n_levels - 1, np.uint8)
        self._check(self.lib.hlala_graph_get_gap_stretch(self.h, a.ctypes.data_as(c_u8p)), "hlala_graph_get_gap_stretch")
        return a

    def batch(self, batch_in: dict) -> "Batch":
        s, keep = fill_struct(BatchIn, batch_in)
        b = C.c_void_p()
        self._check(self.lib.hlala_batch_create(self.h, C.byref(s), C.byref(b)), "hlala_batch_create")
        return Batch(self, b, batch_in["n_chains"], batch_in["n_pairs"])

    def batch_window(self, seeds: "SeedBatch", first_unit: int, n_units: int) -> "Batch":
        """units [first_unit, first_unit + n_units) of a decoded sample as a resident batch: the descriptor points into the sample's arrays (no host
        copy), the batch takes chain_off[0] of the window as its first absolute chain number"""
        d = seeds.window(first_unit, n_units)
        b = C.c_void_p()
        f = self.lib.hlala_batch_create_unpaired if seeds.long_read_mode else self.lib.hlala_batch_create
        self._check(f(self.h, C.byref(d), C.byref(b)), "hlala_batch_create (window)")
        bt = Batch(self, b, d.n_chains, n_units)
        if seeds.long_read_mode:
            bt.unpaired = True
        return bt

    def batch_unpaired(self, batch_in: dict) -> "Batch":
        """Long-read / unpaired mode: batch_in["n_pairs"] is the number of reads (hlala_batch_create_unpaired)."""
        s, keep = fill_struct(BatchIn, batch_in)
        b = C.c_void_p()
        self._check(self.lib.hlala_batch_create_unpaired(self.h, C.byref(s), C.byref(b)), "hlala_batch_create_unpaired")
        bt = Batch(self, b, batch_in["n_chains"], batch_in["n_pairs"]); bt.unpaired = True
        return bt

    def batch_from_seeds(self, seeds_in: dict) -> "Batch":
        s, keep = fill_struct(SeedsIn, seeds_in)
        b = C.c_void_p()
        self._check(self.lib.hlala_batch_create_from_seeds(self.h, C.byref(s), C.byref(b)), "hlala_batch_create_from_seeds")
        return Batch(self, b, seeds_in["n_chains"], 0)

    def exon_loglik(self, exon_in: dict):
        """HLATyper per-cluster x per-read log-likelihoods and mismatch counts (hlala_exon_loglik)."""
        s, keep = fill_struct(ExonIn, exon_in)
        Cn, R = exon_in["n_clusters"], exon_in["n_reads"]
        LL = np.zeros(Cn * R, np.float64); mism = np.zeros(Cn * R, np.int32)
        self._check(self.lib.hlala_exon_loglik(self.h, C.byref(s), LL.ctypes.data_as(c_f64p), mism.ctypes.data_as(c_i32p)), "hlala_exon_loglik")
        return LL.reshape(Cn, R), mism.reshape(Cn, R)

    def pair_loglik(self, LL, mism):
        """All cluster pairs c1 <= c2 in the reference's single-thread order (hlala_pair_loglik)."""
        LL = np.ascontiguousarray(LL, np.float64); mism = np.ascontiguousarray(mism, np.int32)
        Cn, R = LL.shape
        n = Cn * (Cn + 1) // 2
        out = [np.zeros(n, np.float64) for _ in range(3)]
        self._check(self.lib.hlala_pair_loglik(self.h, LL.ctypes.data_as(c_f64p), mism.ctypes.data_as(c_i32p), Cn, R,
                                               *[o.ctypes.data_as(c_f64p) for o in out]), "hlala_pair_loglik")
        return out

    def call_locus(self, pairLL, misAvg, misMin):
        """The call of one locus from the all-pairs table (hlala_call_locus; hla/HLATyper.cpp:2366-2541)."""
        a = [np.ascontiguousarray(x, np.float64) for x in (pairLL, misAvg, misMin)]
        nP = len(a[0]); Cn = int((np.sqrt(8 * nP + 1) - 1) / 2 + 0.5)
        assert Cn * (Cn + 1) // 2 == nP
        order = np.zeros(nP, np.int32); pn = np.zeros(nP, np.float64); marg = np.zeros(Cn, np.float64); out = CallOut()
        self._check(self.lib.hlala_call_locus(self.h, Cn, *[x.ctypes.data_as(c_f64p) for x in a], order.ctypes.data_as(c_i32p),
                                              pn.ctypes.data_as(c_f64p), marg.ctypes.data_as(c_f64p), C.byref(out)), "hlala_call_locus")
        return dict(order=order, p_normalized=pn, cluster_marginal=marg, first_cluster=out.first_cluster, second_cluster=out.second_cluster,
                    first_marginal=out.first_marginal, second_p=out.second_p, ll_max=out.ll_max, max_pair=out.max_pair, n_sort_ties=out.n_sort_ties)

    def type_locus(self, exon_in: dict, want_reads_table=True):
        """hlala_type_locus: exon_loglik -> pair_loglik -> call_locus with the tables left on the device in between."""
        s, keep = fill_struct(ExonIn, exon_in)
        Cn, R = exon_in["n_clusters"], exon_in["n_reads"]
        nP = Cn * (Cn + 1) // 2
        LL = np.zeros(Cn * R, np.float64) if want_reads_table else None; mism = np.zeros(Cn * R, np.int32) if want_reads_table else None
        pl, ma, mm, pn = [np.zeros(nP, np.float64) for _ in range(4)]; order = np.zeros(nP, np.int32); marg = np.zeros(Cn, np.float64); out = CallOut()
        self.lib.hlala_type_locus.argtypes = [C.c_void_p, C.c_void_p, c_f64p, c_i32p, c_f64p, c_f64p, c_f64p, c_i32p, c_f64p, c_f64p, C.c_void_p]
        self._check(self.lib.hlala_type_locus(self.h, C.byref(s), None if LL is None else LL.ctypes.data_as(c_f64p), None if mism is None else mism.ctypes.data_as(c_i32p),
                                              pl.ctypes.data_as(c_f64p), ma.ctypes.data_as(c_f64p), mm.ctypes.data_as(c_f64p), order.ctypes.data_as(c_i32p),
                                              pn.ctypes.data_as(c_f64p), marg.ctypes.data_as(c_f64p), C.byref(out)), "hlala_type_locus")
        return dict(LL=None if LL is None else LL.reshape(Cn, R), mism=None if mism is None else mism.reshape(Cn, R), pairLL=pl, misAvg=ma, misMin=mm, order=order, p_normalized=pn, cluster_marginal=marg,
                    first_cluster=out.first_cluster, second_cluster=out.second_cluster, first_marginal=out.first_marginal, second_p=out.second_p, ll_max=out.ll_max, max_pair=out.max_pair, n_sort_ties=out.n_sort_ties)

    def kmer_presence(self, batch, queries, k=31, pair_mask=None):
        """hlala_kmer_presence: which query k-mers (strings of length k) occur, in canonical form, in the reads of `batch`."""
        n = len(queries)
        buf = "".join(queries).encode(); assert len(buf) == n * k
        present = np.zeros(max(1, n), np.uint8)
        mask = None if pair_mask is None else np.ascontiguousarray(pair_mask, np.uint8)
        self.lib.hlala_kmer_presence.argtypes = [C.c_void_p, C.c_void_p, c_u8p, C.c_int32, C.c_int32, C.c_char_p, c_u8p]
        self._check(self.lib.hlala_kmer_presence(self.h, batch.b, None if mask is None else mask.ctypes.data_as(c_u8p), k, n, buf, present.ctypes.data_as(c_u8p)), "hlala_kmer_presence")
        return present[:n]

    def kmer_keep_reads(self, batch, pair_mask=None):
        """hlala_kmer_keep_reads: the looked-at reads of `batch` stay on the device for kmer_presence_kept; returns the number of reads kept."""
        mask = None if pair_mask is None else np.ascontiguousarray(pair_mask, np.uint8)
        n = C.c_int64(0)
        self.lib.hlala_kmer_keep_reads.argtypes = [C.c_void_p, C.c_void_p, c_u8p, C.POINTER(C.c_int64)]
        self._check(self.lib.hlala_kmer_keep_reads(self.h, batch.b, None if mask is None else mask.ctypes.data_as(c_u8p), C.byref(n)), "hlala_kmer_keep_reads")
        return n.value

    def kmer_presence_kept(self, queries, k=31):
        """hlala_kmer_presence_kept: kmer_presence over the union of the reads kept since the last kmer_forget_reads."""
        n = len(queries)
        buf = "".join(queries).encode(); assert len(buf) == n * k
        present = np.zeros(max(1, n), np.uint8)
        self.lib.hlala_kmer_presence_kept.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_char_p, c_u8p]
        self._check(self.lib.hlala_kmer_presence_kept(self.h, k, n, buf, present.ctypes.data_as(c_u8p)), "hlala_kmer_presence_kept")
        return present[:n]

    def kmer_forget_reads(self):
        self.lib.hlala_kmer_forget_reads.argtypes = [C.c_void_p]; self.lib.hlala_kmer_forget_reads.restype = None
        self.lib.hlala_kmer_forget_reads(self.h)

    def set_tail_pool(self, k: int):
        """hlala_set_tail_pool: the broad / large / in-memory DP classes of up to k consecutive alignments run in one launch per class."""
        self.lib.hlala_set_tail_pool.argtypes = [C.c_void_p, C.c_int]
        self._check(self.lib.hlala_set_tail_pool(self.h, int(k)), "hlala_set_tail_pool")

    def flush(self):
        self.lib.hlala_flush.argtypes = [C.c_void_p]
        self._check(self.lib.hlala_flush(self.h), "hlala_flush")

    def estimate_insert_size(self, batch_in: dict):
        """processBAM::estimateInsertSize on the primaries of `batch_in` (hlala_estimate_insert_size)."""
        s, keep = fill_struct(BatchIn, batch_in)
        o = InsertSizeOut()
        self._check(self.lib.hlala_estimate_insert_size(self.h, C.byref(s), C.byref(o)), "hlala_estimate_insert_size")
        return dict(mean=o.mean, sd=o.sd, n_used=o.n_used, n_skipped=o.n_skipped, total_weight=o.total_weight)

    def set_gene_intervals(self, first_level, last_level):
        """HLATyper::interestingLevels (graphgene_levelBoundaries, hla/HLATyper.cpp:241-252)."""
        f = np.ascontiguousarray(first_level, np.int32); l = np.ascontiguousarray(last_level, np.int32)
        assert f.shape == l.shape
        self._check(self.lib.hlala_set_gene_intervals(self.h, len(f), f.ctypes.data_as(c_i32p), l.ctypes.data_as(c_i32p)), "hlala_set_gene_intervals")

    def coverage(self, reset=False):
        """bases_per_level accumulated by Batch.postprocess() over all batches of this context (reads_per_level.txt)."""
        info = GraphInfo()
        self._check(self.lib.hlala_graph_get_info(self.h, C.byref(info)), "hlala_graph_get_info")
        out = np.zeros(max(1, info.n_levels - 1), np.int32)
        self._check(self.lib.hlala_get_coverage(self.h, out.ctypes.data_as(c_i32p), int(reset)), "hlala_get_coverage")
        return out

    def close(self):
        if getattr(self, "h", None):
            self.lib.hlala_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Comm:
    """hlala_comm: the contexts of one process (one per GPU) with the exchange steps of the path -- gather of the per-pair records, sum of the coverage
    counters -- over RCCL when they sit on different devices (include/hlala_gpu.h)."""

    def __init__(self, ctxs):
        self.ctxs = list(ctxs); self.lib = lib = self.ctxs[0].lib
        lib.hlala_comm_create.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.POINTER(C.c_void_p)]
        lib.hlala_comm_last_error.restype = C.c_char_p; lib.hlala_comm_last_error.argtypes = [C.c_void_p]
        lib.hlala_comm_destroy.argtypes = [C.c_void_p]; lib.hlala_comm_destroy.restype = None
        lib.hlala_comm_uses_rccl.argtypes = [C.c_void_p]
        lib.hlala_gather_pair_records.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.c_void_p, C.c_int64, C.c_void_p]
        lib.hlala_reduce_coverage.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        hs = (C.c_void_p * len(self.ctxs))(*[c.h for c in self.ctxs]); h = C.c_void_p()
        if lib.hlala_comm_create(hs, len(self.ctxs), C.byref(h)) != 0:
            raise RuntimeError("hlala_comm_create: " + (lib.hlala_comm_last_error(None) or b"").decode())
        self.h = h

    def _check(self, rc, what):
        if rc != 0:
            raise RuntimeError(f"{what}: " + (self.lib.hlala_comm_last_error(self.h) or b"").decode())

    @property
    def uses_rccl(self):
        return bool(self.lib.hlala_comm_uses_rccl(self.h))

    def gather_pair_records(self, batches):
        """batches[i]: the Batch of context i or None.  Returns (records [sum of pairs, 8], counts [contexts])."""
        n = len(self.ctxs)
        bs = (C.c_void_p * n)(*[(b.b if b is not None else None) for b in batches])
        total = int(sum(b.n_pairs for b in batches if b is not None))
        out = np.zeros((max(1, total), 8), np.float64); counts = np.zeros(n, np.int64)
        self._check(self.lib.hlala_gather_pair_records(self.h, bs, out.ctypes.data, total, counts.ctypes.data), "hlala_gather_pair_records")
        return out[:total], counts

    def reduce_coverage(self, reset=False):
        info = GraphInfo()
        self.ctxs[0]._check(self.lib.hlala_graph_get_info(self.ctxs[0].h, C.byref(info)), "hlala_graph_get_info")
        out = np.zeros(max(1, info.n_levels - 1), np.int32)
        self._check(self.lib.hlala_reduce_coverage(self.h, out.ctypes.data, int(reset)), "hlala_reduce_coverage")
        return out

    def close(self):
        if getattr(self, "h", None):
            self.lib.hlala_comm_destroy(self.h); self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Batch:
    def __init__(self, ctx: Context, b, n_chains, n_pairs):
        self.ctx, self.b, self.n_chains, self.n_pairs = ctx, b, n_chains, n_pairs

    def project(self):
        self.ctx._check(self.ctx.lib.hlala_project_chains(self.ctx.h, self.b), "hlala_project_chains")

    def extend(self):
        self.ctx._check(self.ctx.lib.hlala_extend_chains(self.ctx.h, self.b), "hlala_extend_chains")

    def pair(self):
        self.ctx._check(self.ctx.lib.hlala_pair_chains(self.ctx.h, self.b), "hlala_pair_chains")

    def align(self):
        self.ctx._check(self.ctx.lib.hlala_align_batch(self.ctx.h, self.b), "hlala_align_batch")

    def set_first_chain(self, first_chain: int):
        """Absolute index of this batch's chain 0 in the caller's numbering: the DPs then draw the random seeds of the unsplit run."""
        self.ctx.lib.hlala_batch_set_first_chain.argtypes = [C.c_void_p, C.c_uint32]
        self.ctx._check(self.ctx.lib.hlala_batch_set_first_chain(self.b, first_chain), "hlala_batch_set_first_chain")

    def chains(self, stage: int) -> dict:
        s, d = alloc_chains_out(self.n_chains, self.ctx.max_columns)
        self.ctx._check(self.ctx.lib.hlala_batch_get_chains(self.ctx.h, self.b, stage, C.byref(s)), "hlala_batch_get_chains")
        return d

    def pairs(self) -> dict:
        s, d = alloc_pairs_out(self.n_pairs, self.ctx.max_columns)
        self.ctx._check(self.ctx.lib.hlala_batch_get_pairs(self.ctx.h, self.b, C.byref(s)), "hlala_batch_get_pairs")
        return d

    def pairs_scalars(self) -> dict:
        """hlala_batch_get_pairs with NULL column pointers: the per-pair / per-mate scalars only."""
        n = self.n_pairs; nr = n * (1 if getattr(self, "unpaired", False) else 2)
        d = dict(pair_status=np.zeros(n, np.int32), best_chain=np.zeros(nr, np.int32), n_combinations=np.zeros(n, np.int32), pair_ll=np.zeros(n), pair_mapq=np.zeros(n),
                 mate_mapq=np.zeros(nr), strands_valid=np.zeros(n, np.uint8))
        o = PairsOut(); types = dict(PairsOut._fields_)
        for k, v in d.items():
            setattr(o, k, v.ctypes.data_as(types[k]))
        self.ctx._check(self.ctx.lib.hlala_batch_get_pairs(self.ctx.h, self.b, C.byref(o)), "hlala_batch_get_pairs")
        return d

    def export_pair_records(self, device_ptr: int):
        """Write 8 doubles per pair into a device buffer (e.g. a torch tensor's data_ptr())."""
        self.ctx._check(self.ctx.lib.hlala_batch_export_pair_records(self.ctx.h, self.b, C.c_void_p(device_ptr)),
                        "hlala_batch_export_pair_records")

    def postprocess(self):
        """Per-pair post-processing (processBAM.cpp:2411-2446): adds to the context's coverage counters, returns includeInHLA per pair."""
        inc = np.zeros(self.n_pairs, np.uint8)
        self.ctx._check(self.ctx.lib.hlala_postprocess_pairs(self.ctx.h, self.b, inc.ctypes.data_as(c_u8p)), "hlala_postprocess_pairs")
        return inc

    def pairs_packed(self):
        """hlala_batch_get_pairs_packed: the columns of the selected alignments without padding (col_off + concatenated arrays)."""
        nr = self.n_pairs * (1 if getattr(self, "unpaired", False) else 2)
        off = np.zeros(nr + 1, np.int64)
        o = PairsPackedOut(); o.cap_cols = 0; o.col_off = off.ctypes.data_as(c_i64p)
        self.ctx.lib.hlala_batch_get_pairs_packed.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(PairsPackedOut)]
        rc = self.ctx.lib.hlala_batch_get_pairs_packed(self.ctx.h, self.b, C.byref(o))          # sizing call
        if rc not in (0, -4):
            self.ctx._check(rc, "hlala_batch_get_pairs_packed")
        T = int(o.n_cols_total)
        d = dict(col_level=np.zeros(T, np.int32), col_edge=np.zeros(T, np.int32), col_gchar=np.zeros(T, np.uint8), col_schar=np.zeros(T, np.uint8),
                 col_fromseed=np.zeros(T, np.uint8), col_mapq=np.zeros(T, np.uint8))
        o.cap_cols = T
        types = dict(PairsPackedOut._fields_)
        for k, v in d.items():
            setattr(o, k, v.ctypes.data_as(types[k]))
        self.ctx._check(self.ctx.lib.hlala_batch_get_pairs_packed(self.ctx.h, self.b, C.byref(o)), "hlala_batch_get_pairs_packed")
        d["col_off"] = off; d["n_cols_total"] = T
        return d

    def unit_stats(self):
        """hlala_unit_alignment_stats: per pair / read strands, distance, fraction OK, weighted OK, columns and mapping qualities."""
        n = self.n_pairs
        d = dict(valid=np.zeros(n, np.uint8), strands_valid=np.zeros(n, np.uint8), distance=np.zeros(n, np.int32), fraction_ok=np.zeros(2 * n), weighted_ok=np.zeros(2 * n),
                 n_columns=np.zeros(2 * n, np.int32), mate_mapq=np.zeros(2 * n))
        o, keep = unit_stats_struct(d)
        d = dict(zip([k for k, _ in UnitStatsOut._fields_], keep))
        self.ctx.lib.hlala_unit_alignment_stats.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(UnitStatsOut)]
        self.ctx._check(self.ctx.lib.hlala_unit_alignment_stats(self.ctx.h, self.b, C.byref(o)), "hlala_unit_alignment_stats")
        return d

    def exon_positions(self, level_min, level_to_exon, insert_mean, insert_sd, min_mapq=0.0, min_weighted_ok=0.0, pair_mask=None, min_alignment_columns=1000):
        """Exon positions of this batch's read pairs for one locus (hlala_exon_positions; hla/HLATyper.cpp:1385-1428)."""
        L, keep = make_locus_desc(level_min, level_to_exon, insert_mean, insert_sd, min_mapq, min_weighted_ok, pair_mask, min_alignment_columns)
        o, d = alloc_exon_positions_out(0, 0, 0)
        rc = self.ctx.lib.hlala_exon_positions(self.ctx.h, self.b, C.byref(L), C.byref(o))          # sizing call
        if rc not in (0, E_CAPACITY):
            self.ctx._check(rc, "hlala_exon_positions")
        o, d = alloc_exon_positions_out(o.n_reads, o.n_pos, o.n_chars)
        self.ctx._check(self.ctx.lib.hlala_exon_positions(self.ctx.h, self.b, C.byref(L), C.byref(o)), "hlala_exon_positions")
        return trim_exon_positions(o, d)

    def work_counters(self):
        """B.work_counter of the batch's last stages (diagnostics; indices: DEBUG_WC_* = include/hlala_gpu.h's HLALA_DEBUG_WC_*)."""
        wc = (C.c_int * DEBUG_WC_N)()
        self.ctx.lib.hlala_debug_work_counters.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_int), C.c_int]
        self.ctx._check(self.ctx.lib.hlala_debug_work_counters(self.ctx.h, self.b, wc, DEBUG_WC_N), "hlala_debug_work_counters")
        return np.array(wc[:], np.int64)

    def dp_items(self):
        """(items [2 * n_chains, 8], retry lists [16 * n_chains]) of the batch's last extension stage (hlala_debug_dp_items)."""
        nc = int(self.n_chains)
        items = np.zeros((2 * nc, 8), np.int32); retry = np.zeros(16 * nc, np.int32)
        self.ctx.lib.hlala_debug_dp_items.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_longlong, C.c_void_p, C.c_longlong]
        self.ctx._check(self.ctx.lib.hlala_debug_dp_items(self.ctx.h, self.b, items.ctypes.data, items.nbytes, retry.ctypes.data, retry.nbytes), "hlala_debug_dp_items")
        return items, retry

    def stats(self) -> BatchStats:
        st = BatchStats()
        self.ctx._check(self.ctx.lib.hlala_batch_get_stats(self.ctx.h, self.b, C.byref(st)), "hlala_batch_get_stats")
        return st

    def close(self):
        if getattr(self, "b", None):
            self.ctx.lib.hlala_batch_destroy(self.b)
            self.b = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
