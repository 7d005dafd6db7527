"""BAM file + graph directory in, result files out: every stage of the path through the C ABI in the order HLA-LA runs them
(mapper/processBAM.cpp:1789-1870 alignReads..., hla/HLATyper.cpp:934-2810 HLATypeInference), on a sample simulated from two known alleles."""
import ctypes as C

import numpy as np
import pytest

from test_bam import batch_records, write_bam
from tools import synth

pytestmark = pytest.mark.gpu


def write_graph_dir(root, H, exons):
    """PRG/ for a world whose graph levels are the columns of H: padding segments and one `gene` segment per exon of locus A."""
    prg = root / "PRG"; prg.mkdir(parents=True)
    G = H.shape[1]; names = []; cuts = [0]
    for a, b in exons:
        cuts += [a, b]
    cuts.append(G)
    k = 0
    for i in range(len(cuts) - 1):
        a, b = cuts[i], cuts[i + 1]
        if a == b:
            continue
        k += 1
        is_exon = (a, b) in exons
        fn = "%d_gene_HLA-A_%d_exon_%d.txt" % (k, k, 2 + exons.index((a, b))) if is_exon else "%d_pad_%d.txt" % (k, k)
        lines = ["IndividualID " + " ".join("L%d" % x for x in range(a, b))]
        if is_exon:
            for h in range(H.shape[0]):
                lines.append("A*%02d:01 " % (h + 1) + " ".join(chr(c) for c in H[h, a:b]))
        else:
            lines.append("ref " + " ".join(chr(c) for c in H[0, a:b]))
        (prg / fn).write_text("\n".join(lines) + "\n"); names.append(fn)
    (prg / "segments.txt").write_text("\n".join(names) + "\n")


def test_bam_and_graph_dir_to_result_files(pkg, tmp_path):
    G = 4000; exons = [(1200, 1470), (1900, 2176)]
    w = synth.make_world(seed=12, G=G, k=1, n_mut=6, mut_density=0.03)
    H = w["H"]; truth = (2, 5)
    lib = C.CDLL(pkg.LIB_PATH)
    write_graph_dir(tmp_path, H, exons)
    T = pkg.Typer(lib, tmp_path); L = T.locus("A")
    assert T.level_names() == ["L%d" % i for i in range(G)] and L.n_columns == 270 + 276
    want = {L.type_cluster("A*%02d:01" % (h + 1)) for h in truth}
    assert len(want) == 2
    # ---- reads of a heterozygous sample, as a BAM file against the haplotype contigs
    b = synth.make_batch(w, 700, seed=77, haps=truth)
    clen = np.diff(w["contigs"]["contig_off"]); nct = w["contigs"]["n_contigs"]
    refs = [("hap%d" % i, int(clen[i])) for i in range(nct)]
    bam = tmp_path / "sample.bam"
    write_bam(bam, refs, batch_records(b, np.random.default_rng(1)), block=30000)
    seeds, names, cnt = pkg.bam_extract_seeds(lib, bam, [("hap%d" % i, 0, int(clen[i]) - 1, i) for i in range(nct)])
    assert seeds["n_pairs"] == 700
    # ---- insert size from the sample, alignment, post-processing
    ctx0 = pkg.Context(w["graph"], w["contigs"], insert_mean=200.0, insert_sd=35.0, rng_seed=5)
    ins = ctx0.estimate_insert_size(seeds)
    assert abs(ins["mean"] - b["insert_mean"]) < 25
    ctx = pkg.Context(w["graph"], w["contigs"], insert_mean=ins["mean"], insert_sd=ins["sd"], rng_seed=5)
    gb = ctx.batch(seeds); gb.align()
    assert gb.stats().n_errors == 0
    genes = T.genes(); assert genes == [("HLA-A", 1200, 2175)]
    ctx.set_gene_intervals([g[1] for g in genes], [g[2] for g in genes])
    include = gb.postprocess()
    assert 50 < include.sum() < 700
    # ---- locus A: exon positions -> filters -> likelihoods -> all pairs -> call
    e = gb.exon_positions(L.level_min, L.level_to_exon, ins["mean"], ins["sd"], pair_mask=include)
    prm = pkg.default_filter_params(first20_n=6)
    use, ignored, fst = pkg.filter_positions(lib, e, prm)
    xin = pkg.exon_in_from_positions(e, use, L.cluster_seq, L.n_clusters, L.n_columns)
    LL, M = ctx.exon_loglik(xin)
    pair_ll, mis_avg, mis_min = ctx.pair_loglik(LL, M)
    call = ctx.call_locus(pair_ll, mis_avg, mis_min)
    assert {call["first_cluster"], call["second_cluster"]} == want
    # ---- k-mer support of the two called alleles in the reads that went into typing
    kc = []
    for c in (call["first_cluster"], call["second_cluster"]):
        q, total = L.cluster_kmers(c, 31)
        kc.append(-1.0 if total == 0 else float(ctx.kmer_presence(gb, q, 31, include).sum()) / total)
    assert min(kc) > 0.9
    # ---- files
    out = tmp_path / "hla"
    pkg.typer_begin_output(lib, out)
    co = pkg.CallOut(call["first_cluster"], call["second_cluster"], call["first_marginal"], call["second_p"], call["ll_max"], call["max_pair"], call["n_sort_ties"])
    # per-pair statistics: against the columns of the chosen alignments (alignmentFractionOK, pairsDistanceInGraphLevels) and the exon positions
    us = gb.unit_stats(); pr = gb.pairs(); stride = ctx.max_columns
    assert us["valid"].all()
    for p in range(0, 700, 7):
        fl = []
        for m in range(2):
            r = 2 * p + m; n = pr["n_cols"][r]; gch = pr["col_gchar"][r * stride:r * stride + n]; sch = pr["col_schar"][r * stride:r * stride + n]
            both = (gch == ord("_")) & (sch == ord("_"))
            assert us["fraction_ok"][r] == ((gch == sch) & ~both).sum() / (~both).sum() and us["n_columns"][r] == n
            lv = pr["col_level"][r * stride:r * stride + n]; lv = lv[lv != -1]; fl.append((int(lv[0]), int(lv[-1])))
        d = fl[1][0] - fl[0][1] - 1 if fl[0][0] < fl[1][0] else fl[0][0] - fl[1][1] - 1
        assert us["distance"][p] == d and us["strands_valid"][p] == pr["strands_valid"][p]
    assert np.array_equal(us["weighted_ok"].reshape(-1, 2)[e["read_pair"]].reshape(-1), e["read_weighted_ok"]) and np.array_equal(us["mate_mapq"], pr["mate_mapq"])
    pkg.typer_write_summary(lib, out, us, unit_mask=include, insert_mean=ins["mean"], insert_sd=ins["sd"])
    summ = (out / "summaryStatistics.txt").read_text()
    assert "Total number (paired) alignments:                 %d\n" % include.sum() in summ and "(unpaired) alignments:                 0\n" in summ
    res = L.write_files(out, e, names, names, pair_ll, mis_avg, mis_min, call["order"], call["p_normalized"], co, kmers_covered=kc, params=prm,
                        unit_stats=us, unit_mask=include, insert_mean=ins["mean"], insert_sd=ins["sd"])
    pkg.typer_end_output(lib, out, ["A"])
    hist = (out / "histogram_matchesPerRead.txt").read_text().splitlines()
    assert hist[0] == "Locus\tLevelValue" and sum(l.startswith("A\treadPair") for l in hist) == e["n_pairs_ok"] and sum(l.startswith("A\tbase") for l in hist) == res.n_piled_positions
    rows = [r.split("\t") for r in (out / "R1_bestguess.txt").read_text().splitlines()]
    assert len(rows) == 3 and {rows[1][2], rows[2][2]} == {L.cluster_id(c) for c in want} and rows[1][0] == "A" and rows[1][1] == "1" and rows[2][1] == "2"
    assert float(rows[1][5]) > 10 and res.locus_coverage > 10 and res.minimum_coverage > 0 and res.avg_column_error < 0.02
    pile = (out / "R1_pileup_A.txt").read_text().splitlines()
    assert len(pile) == 546 and pile[0].split("\t")[:2] == ["0", "0"] and pile[270].split("\t")[:2] == ["1", "0"]
    ids = (out / "R1_readIDs_A.txt").read_text().split()
    assert len(ids) == res.n_utilized_reads and set(ids) <= set(names)
    npairs = L.n_clusters * (L.n_clusters + 1) // 2
    assert len((out / "R1_PP_A_pairs.txt").read_text().splitlines()) == npairs + 1
    assert len((out / "R1_columnIncompatibilities_A.txt").read_text().splitlines()) == 547


def test_graph_directory_files_align_like_the_oracle(pkg, oracle, tmp_path):
    """PRG/graph.txt, sequences.txt, the reference FASTA, translation files and a BAM -> context and batch built from files only.
    The contigs carry the extra level-0 position of translation files ending in a newline (mapper/processBAM.cpp:4406-4412 with
    Utilities::StrtoI("") == 0); product and oracle get the same arrays, and pairs placed away from level 0 come out as for the source arrays."""
    from test_gpu_align import assert_pairs_equal, run_both
    from test_graph_files import write_contigs_dir, write_graph_txt
    from util import compare_chains
    w = synth.make_world(seed=21, G=3000, k=2)
    b = synth.make_batch(w, 300, seed=22)
    lib = C.CDLL(pkg.LIB_PATH)
    (tmp_path / "PRG").mkdir()
    write_graph_txt(tmp_path / "PRG" / "graph.txt", w["graph"], np.random.default_rng(2))
    write_contigs_dir(tmp_path, w["contigs"], np.random.default_rng(3))
    graph = pkg.load_graph_text(lib, tmp_path / "PRG" / "graph.txt")
    contigs, intervals = pkg.load_contigs_dir(lib, tmp_path, extended_reference_genome=False)
    clen = np.diff(w["contigs"]["contig_off"])
    assert np.array_equal(np.diff(contigs["contig_off"]), clen + 1)                 # every translation file ended in a newline
    refs = [(iv[0], int(clen[i])) for i, iv in enumerate(intervals)]
    bam = tmp_path / "s.bam"
    write_bam(bam, refs, batch_records(b, np.random.default_rng(4)), block=25000)
    seeds, names, cnt = pkg.bam_extract_seeds(lib, bam, intervals)
    assert seeds["n_pairs"] == 300 and cnt["incomplete"] == 0
    seeds["insert_mean"], seeds["insert_sd"] = b["insert_mean"], b["insert_sd"]
    exp, gb, ctx = run_both(pkg, oracle, dict(graph=graph, contigs=contigs), seeds)
    compare_chains(gb.chains(1), exp["ext"], seeds["n_chains"], label="from files")
    assert_pairs_equal(gb.pairs(), exp["pairs"])
    # the same batch against the source arrays: identical except where the level-0 quirk can reach (pairs next to level 0)
    exp0, gb0, ctx0 = run_both(pkg, oracle, w, seeds)
    p1, p0 = gb.pairs(), gb0.pairs()
    far = (p0["n_cols"].reshape(-1, 2).min(1) > 0) & (p0["col_level"].reshape(300, -1).max(1) > 50)
    for k in ("best_chain", "n_cols", "col_level", "col_mapq"):
        a1 = p1[k].reshape(300, -1); a0 = p0[k].reshape(300, -1)
        assert np.array_equal(a1[far], a0[far]), k
    assert far.sum() > 250


def test_cpp_host_mirror_from_files_to_result_files(pkg, tmp_path):
    """hla-la_amd/host/hlala_host.hpp (processBAM + HLATyper with the reference's names) driven by a C++ program: the files it writes
    are the files the ctypes path writes from the same inputs."""
    import os
    import subprocess
    from test_graph_files import write_contigs_dir, write_graph_txt
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    G = 4000; exons = [(1200, 1470), (1900, 2176)]
    w = synth.make_world(seed=12, G=G, k=1, n_mut=6, mut_density=0.03)
    lib = C.CDLL(pkg.LIB_PATH)
    write_graph_dir(tmp_path, w["H"], exons)
    write_graph_txt(tmp_path / "PRG" / "graph.txt", w["graph"], np.random.default_rng(2))
    write_contigs_dir(tmp_path, w["contigs"], np.random.default_rng(3))
    b = synth.make_batch(w, 700, seed=77, haps=(2, 5))
    contigs, intervals = pkg.load_contigs_dir(lib, tmp_path, extended_reference_genome=False)
    clen = np.diff(w["contigs"]["contig_off"])
    bam = tmp_path / "sample.bam"
    write_bam(bam, [(iv[0], int(clen[i])) for i, iv in enumerate(intervals)], batch_records(b, np.random.default_rng(1)), block=30000)
    # ---- C++
    exe = str(tmp_path / "typer_mirror")
    subprocess.check_call(["g++", "-std=c++11", "-O1", "-Wall", "-o", exe, os.path.join(ROOT, "tests", "host_cpp", "test_typer_mirror.cpp"),
                           "-L", os.path.join(ROOT, "hla-la_amd"), "-lhlala_gpu", "-Wl,-rpath," + os.path.join(ROOT, "hla-la_amd")])
    out_cpp = tmp_path / "out_cpp"
    r = subprocess.run([exe, str(tmp_path), str(bam), str(out_cpp), "A"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "TYPED A A*03:01 A*06:01" in r.stdout or "TYPED A A*06:01 A*03:01" in r.stdout, r.stdout + r.stderr
    # ---- the same through the ctypes binding
    graph = pkg.load_graph_text(lib, tmp_path / "PRG" / "graph.txt")
    seeds, names, cnt = pkg.bam_extract_seeds(lib, bam, intervals)
    ctx0 = pkg.Context(graph, contigs, insert_mean=200.0, insert_sd=35.0, rng_seed=5)
    ins = ctx0.estimate_insert_size(seeds)
    ctx = pkg.Context(graph, contigs, insert_mean=ins["mean"], insert_sd=ins["sd"], rng_seed=5)
    gb = ctx.batch(seeds); gb.align()
    T = pkg.Typer(lib, tmp_path); L = T.locus("A")
    genes = T.genes(); ctx.set_gene_intervals([g[1] for g in genes], [g[2] for g in genes])
    include = gb.postprocess()
    e = gb.exon_positions(L.level_min, L.level_to_exon, ins["mean"], ins["sd"], pair_mask=include)
    prm = pkg.default_filter_params(first20_n=6)
    use, ignored, fst = pkg.filter_positions(lib, e, prm)
    LL, M = ctx.exon_loglik(pkg.exon_in_from_positions(e, use, L.cluster_seq, L.n_clusters, L.n_columns))
    pair_ll, mis_avg, mis_min = ctx.pair_loglik(LL, M)
    call = ctx.call_locus(pair_ll, mis_avg, mis_min)
    kc = []
    for c in (call["first_cluster"], call["second_cluster"]):
        q, total = L.cluster_kmers(c, 31)
        kc.append(-1.0 if total == 0 else float(ctx.kmer_presence(gb, q, 31, include).sum()) / total)
    out_py = tmp_path / "out_py"
    pkg.typer_begin_output(lib, out_py)
    co = pkg.CallOut(call["first_cluster"], call["second_cluster"], call["first_marginal"], call["second_p"], call["ll_max"], call["max_pair"], call["n_sort_ties"])
    us = gb.unit_stats()
    pkg.typer_write_summary(lib, out_py, us, unit_mask=include, insert_mean=ins["mean"], insert_sd=ins["sd"])
    L.write_files(out_py, e, names, names, pair_ll, mis_avg, mis_min, call["order"], call["p_normalized"], co, kmers_covered=kc, params=prm,
                  unit_stats=us, unit_mask=include, insert_mean=ins["mean"], insert_sd=ins["sd"])
    pkg.typer_end_output(lib, out_py, ["A"])
    files = sorted(os.listdir(out_py))
    assert files == sorted(os.listdir(out_cpp)) and "R1_bestguess.txt" in files and "summaryStatistics.txt" in files and len(files) == 9
    for fn in files:
        assert (out_py / fn).read_bytes() == (out_cpp / fn).read_bytes(), fn
    assert ("insert size %.3f %.3f pairs 700" % (ins["mean"], ins["sd"])) in r.stdout


def test_long_reads_from_bam_to_result_files(pkg, tmp_path):
    """BASELINE config 5 style: unpaired kilobase reads with ONT-like errors; BAM (long-read mode: primaries only) -> unpaired batch ->
    per-read exon positions -> filters incl. the strand filter -> likelihoods -> call -> files."""
    G = 6000; exons = [(2200, 2470), (3100, 3376)]
    w = synth.make_world(seed=14, G=G, k=1, n_mut=6, mut_density=0.03)
    truth = (1, 4)
    lib = C.CDLL(pkg.LIB_PATH)
    write_graph_dir(tmp_path, w["H"], exons)
    T = pkg.Typer(lib, tmp_path); L = T.locus("A")
    want = {L.type_cluster("A*%02d:01" % (h + 1)) for h in truth}
    assert len(want) == 2
    u = synth.make_long_batch(w, 160, seed=31, len_lo=1500, len_hi=3000, haps=truth, p_second=0.3)
    clen = np.diff(w["contigs"]["contig_off"]); nct = w["contigs"]["n_contigs"]
    recs = []
    for r in range(u["n_pairs"]):
        seq = bytes(u["read_bases"][u["read_off"][r]:u["read_off"][r + 1]]).decode(); qual = (u["read_quals"][u["read_off"][r]:u["read_off"][r + 1]].astype(int) - 33).tolist()
        for c in range(u["chain_off"][r], u["chain_off"][r + 1]):
            cig = [(int(x) >> 4, "MIDNSHP=X"[int(x) & 15]) for x in u["cigar"][u["cigar_off"][c]:u["cigar_off"][c + 1]]]
            recs.append(dict(name="long%05d" % r, flag=(16 if u["chain_reverse"][c] else 0) | (0 if c == u["read_primary"][r] else 256), ref=int(u["chain_contig"][c]),
                             pos=int(u["chain_pos"][c]), cigar=cig, seq=seq, qual=qual, tags=[("AS", "i", int(u["chain_as"][c]))]))
    bam = tmp_path / "long.bam"
    write_bam(bam, [("hap%d" % i, int(clen[i])) for i in range(nct)], recs, block=40000)
    seeds, names, cnt = pkg.bam_extract_seeds(lib, bam, [("hap%d" % i, 0, int(clen[i]) - 1, i) for i in range(nct)], long_read_mode=True)
    assert seeds["n_pairs"] == 160 and seeds["n_chains"] == 160                       # secondaries are not taken in long-read mode
    ctx = pkg.Context(w["graph"], w["contigs"], insert_mean=200.0, insert_sd=35.0, rng_seed=5, long_read_mode=1, max_columns=8192)
    gb = ctx.batch_unpaired(seeds); gb.align()
    assert gb.stats().n_errors == 0
    genes = T.genes(); ctx.set_gene_intervals([g[1] for g in genes], [g[2] for g in genes])
    include = gb.postprocess()
    assert 20 < include.sum() <= 160
    e = gb.exon_positions(L.level_min, L.level_to_exon, 0, 0, pair_mask=include, min_alignment_columns=1000)
    assert e["n_reads"] > 20 and (e["read_mapq"][1::2] == -1).all()
    prm = pkg.default_filter_params(first20_n=6, long_read_strand_filter=1, strand_min_allele_coverage=8, strand_min_freq=0.1)
    use, ignored, fst = pkg.filter_positions(lib, e, prm)
    LL, M = ctx.exon_loglik(pkg.exon_in_from_positions(e, use, L.cluster_seq, L.n_clusters, L.n_columns))
    pair_ll, mis_avg, mis_min = ctx.pair_loglik(LL, M)
    call = ctx.call_locus(pair_ll, mis_avg, mis_min)
    assert {call["first_cluster"], call["second_cluster"]} == want
    out = tmp_path / "hla"
    pkg.typer_begin_output(lib, out)
    co = pkg.CallOut(call["first_cluster"], call["second_cluster"], call["first_marginal"], call["second_p"], call["ll_max"], call["max_pair"], call["n_sort_ties"])
    res = L.write_files(out, e, names, None, pair_ll, mis_avg, mis_min, call["order"], call["p_normalized"], co, params=prm, long_read_mode=True)
    rows = [r.split("\t") for r in (out / "R1_bestguess.txt").read_text().splitlines()]
    assert {rows[1][2], rows[2][2]} == {L.cluster_id(c) for c in want} and res.n_utilized_reads > 5      # error-rich reads: the first-N filter drops many of them
    pile = (out / "R1_pileup_A.txt").read_text().splitlines()
    assert len(pile) == 546 and all(" long" in l for l in pile if int(l.split("\t")[2]) > 0)
