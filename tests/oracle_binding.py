"""ctypes binding of oracle/_build/liboracle.so -- the CPU checker.  Tests / smoke / bench cpu_baseline only."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
from conftest import load_package  # noqa: E402

P = load_package()
_so = os.path.join(ROOT, "oracle", "_build", "liboracle.so")
_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(_so)
        vp = C.c_void_p
        _lib.orc_create.argtypes = [C.POINTER(P.GraphDesc), C.POINTER(P.ContigsDesc), C.POINTER(P.Params)]
        _lib.orc_create.restype = vp
        _lib.orc_destroy.argtypes = [vp]
        _lib.orc_last_error.restype = C.c_char_p
        _lib.orc_graph_info.argtypes = [vp, C.POINTER(P.GraphInfo)]
        _lib.orc_graph_get_paths.argtypes = [vp, P.c_i32p, P.c_i32p, P.c_i32p]
        _lib.orc_graph_get_gap_stretch.argtypes = [vp, P.c_u8p]
        _lib.orc_extend_seeds.argtypes = [vp, C.POINTER(P.SeedsIn), C.POINTER(P.ChainsOut), P.c_i64p]
        _lib.orc_align_batch.argtypes = [vp, C.POINTER(P.BatchIn), C.POINTER(P.ChainsOut), C.POINTER(P.ChainsOut),
                                         C.POINTER(P.PairsOut), C.c_int, P.c_i64p]
        _lib.orc_rethread_columns.argtypes = [vp, C.POINTER(P.SeedsIn), C.c_int, C.POINTER(P.ChainsOut)]
        _lib.orc_intervals_overlap.argtypes = [C.c_int] * 4
        _lib.orc_phred.argtypes = [C.c_int, P.c_f64p, P.c_u8p, P.c_u8p, P.c_f64p]
        _lib.orc_rand_r.argtypes = [C.c_int, P.c_u32p, P.c_i32p]
        _lib.orc_exon_loglik.argtypes = [C.POINTER(P.ExonIn), C.c_int, P.c_f64p, P.c_i32p]
        _lib.orc_pair_loglik.argtypes = [P.c_f64p, P.c_i32p, C.c_int, C.c_int, P.c_f64p, P.c_f64p, P.c_f64p]
        _lib.orc_normal_logpdf_penalty.argtypes = [C.c_double, C.c_double]
        _lib.orc_normal_logpdf_penalty.restype = C.c_double
        _lib.orc_align_long_reads.argtypes = [vp, C.POINTER(P.BatchIn), C.POINTER(P.ChainsOut), C.POINTER(P.ChainsOut), C.POINTER(P.PairsOut)]
        _lib.orc_estimate_insert_size.argtypes = [vp, C.POINTER(P.BatchIn), C.POINTER(P.InsertSizeOut)]
        _lib.orc_insert_size_from_histogram.argtypes = [C.c_int, P.c_i32p, P.c_f64p, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    return _lib


class OracleError(RuntimeError):
    pass


def exon_loglik(exon_in, long_read_mode=0):
    s, keep = P.fill_struct(P.ExonIn, exon_in)
    Cn, R = exon_in["n_clusters"], exon_in["n_reads"]
    LL = np.zeros(Cn * R, np.float64); mism = np.zeros(Cn * R, np.int32)
    lib().orc_exon_loglik(C.byref(s), long_read_mode, LL.ctypes.data_as(P.c_f64p), mism.ctypes.data_as(P.c_i32p))
    return LL.reshape(Cn, R), mism.reshape(Cn, R)


def pair_loglik(LL, mism):
    LL = np.ascontiguousarray(LL, np.float64); mism = np.ascontiguousarray(mism, np.int32)
    Cn, R = LL.shape
    out = [np.zeros(Cn * (Cn + 1) // 2, np.float64) for _ in range(3)]
    lib().orc_pair_loglik(LL.ctypes.data_as(P.c_f64p), mism.ctypes.data_as(P.c_i32p), Cn, R, *[o.ctypes.data_as(P.c_f64p) for o in out])
    return out


class Oracle:
    def __init__(self, graph, contigs, insert_mean=200.0, insert_sd=35.0, rng_seed=12345, long_read_mode=0, max_columns=384):
        L = lib()
        self.max_columns = max_columns
        g, self._kg = P.fill_struct(P.GraphDesc, graph)
        params = P.Params(insert_mean, insert_sd, rng_seed, long_read_mode, max_columns, 0)
        cp = None
        if contigs is not None:
            c, self._kc = P.fill_struct(P.ContigsDesc, contigs)
            cp = C.byref(c)
        self.h = L.orc_create(C.byref(g), cp, C.byref(params))
        if not self.h:
            raise OracleError(L.orc_last_error().decode())

    def _check(self, rc):
        if rc != 0:
            raise OracleError(lib().orc_last_error().decode())

    def dp_maxima(self, reset=False):
        """Largest frontier, candidate-cell set, kept-cell table and sequence-complete set of any DP call so far."""
        a = np.zeros(4, np.int64)
        lib().orc_dp_maxima.argtypes = [C.c_void_p, P.c_i64p, C.c_int]
        lib().orc_dp_maxima(self.h, a.ctypes.data_as(P.c_i64p), int(reset))
        return dict(frontier=int(a[0]), targets=int(a[1]), kept_cells=int(a[2]), completed=int(a[3]))

    def dp_histogram(self, reset=False):
        """Per DP call of the calling thread: counts by ceil(log2) of the widest frontier / of the largest target set (bucket b = (2^(b-1), 2^b])."""
        a = np.zeros(32, np.int64)
        lib().orc_dp_histogram.argtypes = [C.c_void_p, P.c_i64p, C.c_int]
        lib().orc_dp_histogram(self.h, a.ctypes.data_as(P.c_i64p), int(reset))
        return dict(frontier=a[:16].copy(), targets=a[16:].copy())

    def graph_info(self):
        gi = P.GraphInfo()
        lib().orc_graph_info(self.h, C.byref(gi))
        return gi

    def graph_paths(self):
        gi = self.graph_info()
        a = [np.zeros(gi.n_paths, np.int32) for _ in range(3)]
        lib().orc_graph_get_paths(self.h, *[x.ctypes.data_as(P.c_i32p) for x in a])
        return a

    def graph_gap_stretch(self):
        gi = self.graph_info()
        a = np.zeros(gi.n_levels - 1, np.uint8)
        lib().orc_graph_get_gap_stretch(self.h, a.ctypes.data_as(P.c_u8p))
        return a

    def extend_seeds(self, seeds_in):
        s, keep = P.fill_struct(P.SeedsIn, seeds_in)
        o, d = P.alloc_chains_out(seeds_in["n_chains"], self.max_columns)
        stats = np.zeros(4, np.int64)
        self._check(lib().orc_extend_seeds(self.h, C.byref(s), C.byref(o), stats.ctypes.data_as(P.c_i64p)))
        d["_stats"] = stats
        return d

    def rethread_columns(self, seeds_in, restrict_gaps=False):
        s, keep = P.fill_struct(P.SeedsIn, seeds_in)
        o, d = P.alloc_chains_out(seeds_in["n_chains"], self.max_columns)
        self._check(lib().orc_rethread_columns(self.h, C.byref(s), int(restrict_gaps), C.byref(o)))
        return d

    def align_batch(self, batch_in, stop_after_projection=False):
        s, keep = P.fill_struct(P.BatchIn, batch_in)
        so, sd = P.alloc_chains_out(batch_in["n_chains"], self.max_columns)
        eo, ed = P.alloc_chains_out(batch_in["n_chains"], self.max_columns)
        po, pd = P.alloc_pairs_out(batch_in["n_pairs"], self.max_columns)
        stats = np.zeros(4, np.int64)
        self._check(lib().orc_align_batch(self.h, C.byref(s), C.byref(so), C.byref(eo), C.byref(po),
                                          int(stop_after_projection), stats.ctypes.data_as(P.c_i64p)))
        return dict(seeds=sd, ext=ed, pairs=pd, stats=stats)

    def align_batch_mt(self, batch_in, n_threads=0, pairs_only=False):
        """orc_align_batch_mt: the same result as align_batch, pairs spread over the host cores (OpenMP, dynamic chunks of 64)."""
        s, keep = P.fill_struct(P.BatchIn, batch_in)
        l = lib()
        l.orc_align_batch_mt.argtypes = [C.c_void_p, C.POINTER(P.BatchIn), C.POINTER(P.ChainsOut), C.POINTER(P.ChainsOut), C.POINTER(P.PairsOut),
                                         C.c_int, P.c_i64p, C.POINTER(C.c_int)]
        po, pd = P.alloc_pairs_out(batch_in["n_pairs"], self.max_columns)
        stats = np.zeros(4, np.int64); used = C.c_int(0)
        if pairs_only:
            self._check(l.orc_align_batch_mt(self.h, C.byref(s), None, None, C.byref(po), n_threads, stats.ctypes.data_as(P.c_i64p), C.byref(used)))
            return dict(pairs=pd, stats=stats, threads=used.value)
        so, sd = P.alloc_chains_out(batch_in["n_chains"], self.max_columns)
        eo, ed = P.alloc_chains_out(batch_in["n_chains"], self.max_columns)
        self._check(l.orc_align_batch_mt(self.h, C.byref(s), C.byref(so), C.byref(eo), C.byref(po), n_threads, stats.ctypes.data_as(P.c_i64p), C.byref(used)))
        return dict(seeds=sd, ext=ed, pairs=pd, stats=stats, threads=used.value)

    def align_long_reads(self, batch_in):
        """alignOneLongRead per read of an unpaired batch (n_pairs = number of reads)."""
        s, keep = P.fill_struct(P.BatchIn, batch_in)
        so, sd = P.alloc_chains_out(batch_in["n_chains"], self.max_columns)
        eo, ed = P.alloc_chains_out(batch_in["n_chains"], self.max_columns)
        po, pd = P.alloc_pairs_out(batch_in["n_pairs"], self.max_columns)
        self._check(lib().orc_align_long_reads(self.h, C.byref(s), C.byref(so), C.byref(eo), C.byref(po)))
        return dict(seeds=sd, ext=ed, pairs=pd)

    def estimate_insert_size(self, batch_in):
        s, keep = P.fill_struct(P.BatchIn, batch_in)
        o = P.InsertSizeOut()
        self._check(lib().orc_estimate_insert_size(self.h, C.byref(s), C.byref(o)))
        return dict(mean=o.mean, sd=o.sd, n_used=o.n_used, n_skipped=o.n_skipped, total_weight=o.total_weight)

    def close(self):
        if self.h:
            lib().orc_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def postprocess_pairs(pairs, n_pairs, stride, gene_first, gene_last, n_cov, cov=None):
    """orc_postprocess_pairs on the `pairs` dict of Oracle.align_batch / Batch.pairs(): returns (bases_per_level, includeInHLA)."""
    l = lib()
    gf = np.ascontiguousarray(gene_first, np.int32); gl = np.ascontiguousarray(gene_last, np.int32)
    cov = np.zeros(n_cov, np.int32) if cov is None else cov
    inc = np.zeros(n_pairs, np.uint8)
    st = np.ascontiguousarray(pairs["pair_status"], np.int32); nc = np.ascontiguousarray(pairs["n_cols"], np.int32)
    lv = np.ascontiguousarray(pairs["col_level"], np.int32); g = np.ascontiguousarray(pairs["col_gchar"], np.uint8)
    l.orc_postprocess_pairs.argtypes = [C.c_int, C.c_int, P.c_i32p, P.c_i32p, P.c_i32p, P.c_u8p, C.c_int, P.c_i32p, P.c_i32p, C.c_int, P.c_i32p, P.c_u8p]
    rc = l.orc_postprocess_pairs(n_pairs, stride, st.ctypes.data_as(P.c_i32p), nc.ctypes.data_as(P.c_i32p), lv.ctypes.data_as(P.c_i32p),
                                 g.ctypes.data_as(P.c_u8p), len(gf), gf.ctypes.data_as(P.c_i32p), gl.ctypes.data_as(P.c_i32p), n_cov,
                                 cov.ctypes.data_as(P.c_i32p), inc.ctypes.data_as(P.c_u8p))
    assert rc == 0, rc
    return cov, inc


def call_locus(pairLL, misAvg, misMin):
    """orc_call_locus: the reference's call of one locus (hla/HLATyper.cpp:2366-2541)."""
    l = lib()
    a = [np.ascontiguousarray(x, np.float64) for x in (pairLL, misAvg, misMin)]
    nP = len(a[0]); Cn = int((np.sqrt(8 * nP + 1) - 1) / 2 + 0.5)
    assert Cn * (Cn + 1) // 2 == nP
    order = np.zeros(nP, np.int32); pn = np.zeros(nP, np.float64); marg = np.zeros(Cn, np.float64); out = P.CallOut()
    l.orc_call_locus.argtypes = [C.c_int, P.c_f64p, P.c_f64p, P.c_f64p, P.c_i32p, P.c_f64p, P.c_f64p, C.POINTER(P.CallOut)]
    rc = l.orc_call_locus(Cn, *[x.ctypes.data_as(P.c_f64p) for x in a], order.ctypes.data_as(P.c_i32p), pn.ctypes.data_as(P.c_f64p),
                          marg.ctypes.data_as(P.c_f64p), C.byref(out))
    assert rc == 0, rc
    return dict(order=order, p_normalized=pn, cluster_marginal=marg, first_cluster=out.first_cluster, second_cluster=out.second_cluster,
                first_marginal=out.first_marginal, second_p=out.second_p, ll_max=out.ll_max, max_pair=out.max_pair, n_sort_ties=out.n_sort_ties)


def exon_positions(pairs, batch, stride, level_min, level_to_exon, insert_mean, insert_sd, min_mapq=0.0, min_weighted_ok=0.0, pair_mask=None, unpaired=False, min_alignment_columns=1000):
    """orc_exon_positions on the `pairs` dict of Oracle.align_batch and the reads of `batch` (hla/HLATyper.cpp:1385-1428)."""
    l = lib()
    L, keep = P.make_locus_desc(level_min, level_to_exon, insert_mean, insert_sd, min_mapq, min_weighted_ok, pair_mask, min_alignment_columns)
    n = int(batch["n_pairs"])
    o, d = P.alloc_exon_positions_out(n, 2 * n * stride, 4 * n * stride)
    a = dict(st=np.ascontiguousarray(pairs["pair_status"], np.int32), nc=np.ascontiguousarray(pairs["n_cols"], np.int32),
             lv=np.ascontiguousarray(pairs["col_level"], np.int32), g=np.ascontiguousarray(pairs["col_gchar"], np.uint8),
             s=np.ascontiguousarray(pairs["col_schar"], np.uint8), mq=np.ascontiguousarray(pairs["col_mapq"], np.uint8),
             mm=np.ascontiguousarray(pairs["mate_mapq"], np.float64), sv=np.ascontiguousarray(pairs["strands_valid"], np.uint8),
             ro=np.ascontiguousarray(batch["read_off"], np.int32), rb=np.ascontiguousarray(batch["read_bases"], np.uint8), rq=np.ascontiguousarray(batch["read_quals"], np.uint8))
    l.orc_exon_positions.argtypes = [C.c_int, C.c_int, P.c_i32p, P.c_i32p, P.c_i32p, P.c_u8p, P.c_u8p, P.c_u8p, P.c_f64p, P.c_u8p, P.c_i32p, P.c_u8p, P.c_u8p,
                                     C.POINTER(P.LocusDesc), C.POINTER(P.ExonPositionsOut)]
    if unpaired:
        l.orc_exon_positions_unpaired.argtypes = [C.c_int, C.c_int, P.c_i32p, P.c_i32p, P.c_i32p, P.c_u8p, P.c_u8p, P.c_u8p, P.c_f64p, P.c_i32p, P.c_u8p, P.c_u8p,
                                                  C.POINTER(P.LocusDesc), C.POINTER(P.ExonPositionsOut)]
        rc = l.orc_exon_positions_unpaired(n, stride, a["st"].ctypes.data_as(P.c_i32p), a["nc"].ctypes.data_as(P.c_i32p), a["lv"].ctypes.data_as(P.c_i32p),
                                           a["g"].ctypes.data_as(P.c_u8p), a["s"].ctypes.data_as(P.c_u8p), a["mq"].ctypes.data_as(P.c_u8p), a["mm"].ctypes.data_as(P.c_f64p),
                                           a["ro"].ctypes.data_as(P.c_i32p), a["rb"].ctypes.data_as(P.c_u8p), a["rq"].ctypes.data_as(P.c_u8p), C.byref(L), C.byref(o))
        assert rc == 0, rc
        return P.trim_exon_positions(o, d)
    rc = l.orc_exon_positions(n, stride, a["st"].ctypes.data_as(P.c_i32p), a["nc"].ctypes.data_as(P.c_i32p), a["lv"].ctypes.data_as(P.c_i32p),
                              a["g"].ctypes.data_as(P.c_u8p), a["s"].ctypes.data_as(P.c_u8p), a["mq"].ctypes.data_as(P.c_u8p), a["mm"].ctypes.data_as(P.c_f64p),
                              a["sv"].ctypes.data_as(P.c_u8p), a["ro"].ctypes.data_as(P.c_i32p), a["rb"].ctypes.data_as(P.c_u8p), a["rq"].ctypes.data_as(P.c_u8p),
                              C.byref(L), C.byref(o))
    assert rc == 0, rc
    return P.trim_exon_positions(o, d)


def filter_positions(e, params=None):
    """orc_filter_positions on an exon-positions dict: (pos_use, read_ignored, stats dict)."""
    l = lib()
    o, keep = P.exon_positions_struct(e)
    prm = params or P.default_filter_params()
    use = np.zeros(max(1, o.n_pos), np.uint8); ign = np.zeros(max(1, o.n_reads), np.uint8); st = P.FilterStats()
    l.orc_filter_positions.argtypes = [C.POINTER(P.ExonPositionsOut), C.POINTER(P.FilterParams), P.c_u8p, P.c_u8p, C.POINTER(P.FilterStats)]
    rc = l.orc_filter_positions(C.byref(o), C.byref(prm), use.ctypes.data_as(P.c_u8p), ign.ctypes.data_as(P.c_u8p), C.byref(st))
    assert rc == 0, rc
    return use[:o.n_pos], ign[:o.n_reads], {k: int(getattr(st, k)) for k, _ in P.FilterStats._fields_}
