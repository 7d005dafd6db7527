"""Read / allele filters (hla/HLATyper.cpp:1496-1862, 2102-2120): product host code vs the oracle restatement."""
import ctypes as C

import numpy as np
import pytest

import oracle_binding as ob


def synth_positions(rng, n_reads, n_exon, cov_alleles="ACGT", p_err=0.03, p_ins=0.01, p_lowq=0.05, p_dirty=0.3):
    """Random exon-position lists with the statistics that matter to the filters: most reads are clean (weighted OK exactly 1.0: ties),
    a few carry errors, rare alleles and inserted bases, some positions have a low per-position mapping quality."""
    truth = rng.integers(0, 4, n_exon)
    e = dict(read_pair=[], read_weighted_ok=[], read_fraction_ok=[], read_distance=[], read_cols_nongap=[], pos_off=[0], pos_exon=[], pos_level=[], pos_mate=[],
             pos_mapq=[], pos_novel_gap=[], geno_off=[0], geno_chars=[], qual_chars=[])
    for r in range(n_reads):
        a = int(rng.integers(0, n_exon - 30)); ln = int(rng.integers(20, 120)); b = min(n_exon, a + ln)
        dirty = rng.random() < p_dirty
        w = [1.0, 1.0] if not dirty else [1.0 - rng.integers(1, 6) / 150.0, 1.0 - rng.integers(0, 4) / 150.0]
        e["read_pair"].append(r); e["read_weighted_ok"] += w; e["read_fraction_ok"] += [1.0, 1.0]; e["read_distance"].append(200); e["read_cols_nongap"] += [150, 150]
        for x in range(a, b):
            al = cov_alleles[truth[x]]
            if dirty and rng.random() < p_err * 5:
                al = cov_alleles[int(rng.integers(0, 4))]
            if rng.random() < p_ins:
                al = al + "T"
            if rng.random() < 0.005:
                al = "_"
            e["pos_exon"].append(x); e["pos_level"].append(1000 + x); e["pos_mate"].append(1 + (x - a) * 2 // max(1, b - a))
            e["pos_mapq"].append(ord("#") if rng.random() < p_lowq else ord("I"))          # '#': pCorrect(35 - 33 = 2) = 0.37 < 0.7; 'I': 0.9999
            e["pos_novel_gap"].append(0)
            e["geno_chars"] += list(al.encode()); e["qual_chars"] += [0 if al == "_" else 70] * len(al); e["geno_off"].append(len(e["geno_chars"]))
        e["pos_off"].append(len(e["pos_exon"]))
    dt = dict(read_pair=np.int32, read_weighted_ok=np.float64, read_fraction_ok=np.float64, read_distance=np.int32, read_cols_nongap=np.int32, pos_off=np.int32,
              pos_exon=np.int32, pos_level=np.int32, pos_mate=np.uint8, pos_mapq=np.uint8, pos_novel_gap=np.int32, geno_off=np.int32, geno_chars=np.uint8, qual_chars=np.uint8)
    e = {k: np.array(v, dt[k]) for k, v in e.items()}
    e.update(n_reads=n_reads, n_pos=len(e["pos_exon"]), n_chars=len(e["geno_chars"]))
    return e


def hand_case():
    """22 reads over exon positions 0..2.  Position 0: 21 x 'A' + 1 x 'C' from the read with the lowest weighted OK -> 'C' is not among the first 20."""
    e = dict(read_pair=[], read_weighted_ok=[], read_fraction_ok=[], read_distance=[], read_cols_nongap=[], pos_off=[0], pos_exon=[], pos_level=[], pos_mate=[],
             pos_mapq=[], pos_novel_gap=[], geno_off=[0], geno_chars=[], qual_chars=[])
    for r in range(22):
        w = 1.0 - r * 0.001
        al0 = "C" if r == 21 else "A"
        e["read_pair"].append(r); e["read_weighted_ok"] += [w, w]; e["read_fraction_ok"] += [1, 1]; e["read_distance"].append(0); e["read_cols_nongap"] += [1, 1]
        for x, al in ((0, al0), (1, "G"), (2, "T")):
            if x == 2 and r >= 5:
                continue                                                 # position 2 has only 5 reads: below filterFirst20N, untouched
            e["pos_exon"].append(x); e["pos_level"].append(10 + x); e["pos_mate"].append(1); e["pos_mapq"].append(ord("I")); e["pos_novel_gap"].append(0)
            e["geno_chars"] += list(al.encode()); e["qual_chars"] += [70]; e["geno_off"].append(len(e["geno_chars"]))
        e["pos_off"].append(len(e["pos_exon"]))
    dt = dict(read_pair=np.int32, read_weighted_ok=np.float64, read_fraction_ok=np.float64, read_distance=np.int32, read_cols_nongap=np.int32, pos_off=np.int32,
              pos_exon=np.int32, pos_level=np.int32, pos_mate=np.uint8, pos_mapq=np.uint8, pos_novel_gap=np.int32, geno_off=np.int32, geno_chars=np.uint8, qual_chars=np.uint8)
    e = {k: np.array(v, dt[k]) for k, v in e.items()}
    e.update(n_reads=22, n_pos=len(e["pos_exon"]), n_chars=len(e["geno_chars"]))
    return e


def test_oracle_filters_hand_derived(oracle):
    e = hand_case()
    use, ign, st = ob.filter_positions(e)
    # positions 0 and 1 have 22 entries each (>= 20): considered; position 2 is not.  At position 0 the 'C' of read 21 (worst weighted OK) is kicked out.
    assert st["considered_positions"] == 2 and st["positions_with_removed_alleles"] == 1 and st["considered_alleles"] == 44 and st["removed_alleles"] == 1
    off = e["pos_off"]
    assert use.sum() == e["n_pos"] - 1 and use[off[21]] == 0              # read 21's position 0 ('C') does not enter the likelihood
    assert ign.sum() == 0 and st["reads_kicked_out"] == 0 and st["reads_kicked_out_robust"] == 0     # one kicked position is below the per-read limit of 2
    assert st["bases_used"] == e["n_pos"] - 1
    # without the filter everything with a good per-position quality is used
    from conftest import load_package
    P = load_package()
    use2, _, st2 = ob.filter_positions(e, P.default_filter_params(filter_first20=0))
    assert use2.all() and st2["considered_positions"] == 0
    # the high-coverage filter (settings of :944-946) removes the 1-in-22 allele by frequency instead
    use3, _, st3 = ob.filter_positions(e, P.default_filter_params(filter_first20=0, high_coverage_filter=1, high_coverage_min_coverage=1, high_coverage_min_freq=0.15))
    assert use3.sum() == e["n_pos"] - 1 and use3[off[21]] == 0 and st3["high_coverage_removed_alleles"] == 1


@pytest.mark.parametrize("seed,n_reads,n_exon", [(1, 60, 200), (2, 900, 300), (3, 4000, 546), (4, 2500, 120)])
def test_product_filters_match_oracle(pkg, oracle, seed, n_reads, n_exon):
    rng = np.random.default_rng(seed)
    e = synth_positions(rng, n_reads, n_exon)
    lib = C.CDLL(pkg.LIB_PATH)                    # host code of the product library: no GPU needed
    for prm in (None, pkg.default_filter_params(high_coverage_filter=1, high_coverage_min_coverage=1, high_coverage_min_freq=0.15),
                pkg.default_filter_params(first20_limit_per_read=0), pkg.default_filter_params(min_per_position_mapq=0.2)):
        ue, ie, se = ob.filter_positions(e, prm)
        ug, ig, sg = pkg.filter_positions(lib, e, prm)
        assert np.array_equal(ug, ue) and np.array_equal(ig, ie) and sg == se
    if n_reads >= 900:
        assert se["considered_positions"] > 0 and se["removed_alleles"] > 0


@pytest.mark.parametrize("seed,n_reads,n_exon", [(11, 1500, 150), (12, 4000, 300)])
def test_long_read_strand_filter_matches_oracle(pkg, oracle, seed, n_reads, n_exon):
    """longReads_filterStrand (hla/HLATyper.cpp:1826-1861): alleles seen often enough whose rarer strand is below the minimum are ignored."""
    rng = np.random.default_rng(seed)
    e = synth_positions(rng, n_reads, n_exon)
    # strands: most reads forward with a few reverse, so that common alleles have a thin rarer strand; some reads get balanced strands
    rev = (rng.random(2 * n_reads) < 0.07).astype(np.uint8)
    e["read_reverse"] = rev
    e["read_mapq"] = np.ones(2 * n_reads)
    lib = C.CDLL(pkg.LIB_PATH)
    seen = 0
    for cov, fr in ((100, 0.1), (30, 0.05), (10, 0.2)):
        prm = pkg.default_filter_params(long_read_strand_filter=1, strand_min_allele_coverage=cov, strand_min_freq=fr)
        ue, ie, se = ob.filter_positions(e, prm)
        ug, ig, sg = pkg.filter_positions(lib, e, prm)
        assert np.array_equal(ug, ue) and np.array_equal(ig, ie) and sg == se
        seen += se["strand_removed_alleles"]
        assert se["strand_alleles_enough_coverage"] > 0
    assert seen > 0
    # without the strand array the filter cannot run
    e2 = {k: v for k, v in e.items() if k != "read_reverse"}
    with pytest.raises(Exception):
        pkg.filter_positions(lib, e2, pkg.default_filter_params(long_read_strand_filter=1))
