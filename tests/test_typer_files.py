"""HLATyper host side: graph loci / exon files / allele clusters / G groups and the per-locus result files, against
the reference's rules written out in plain Python (hla/HLATyper.cpp:84-214, 1180-1372, 1883-2044, 2451-2488, 2543-2759, 4086-4207)."""
import ctypes as C
import math
import os

import numpy as np
import pytest

import oracle_binding as ob
from test_filters import synth_positions


def g(v):
    """C++ `ostream << double` with default precision."""
    return "%g" % v


def make_graph_dir(root, rng, n_types=40):
    """PRG/ with segments.txt and segment files in the naming of the reference (`<n>_gene_<locus>_<m>_exon_<N>.txt`)."""
    prg = root / "PRG"; prg.mkdir(parents=True)
    segs = [("1_pad_start.txt", 30, None), ("2_gene_HLA-A_1_intron_1.txt", 25, "A"), ("3_gene_HLA-A_2_exon_2.txt", 60, "A"), ("4_gene_HLA-A_3_intron_2.txt", 20, "A"),
            ("5_gene_HLA-A_4_exon_3.txt", 45, "A"), ("6_pad_mid.txt", 15, None), ("7_gene_DQA1_1_exon_2.txt", 50, "DQA1"), ("8_pad_end.txt", 10, None)]
    (prg / "segments.txt").write_text("\n".join(s[0] for s in segs) + "\n\n")
    types = {"A": ["A*%02d:%02d" % (1 + i // 4, 1 + i % 4) + (":01" if i % 3 == 0 else "") for i in range(n_types)] + ["A_odd_name"],
             "DQA1": ["DQA1*%02d:%02d" % (1 + i // 3, 1 + i % 3) for i in range(12)]}
    level = 0; content = {}
    base = {}
    for name, n, gene in segs:
        cols = ["L%d_%s" % (level + i, name.split("_")[0]) for i in range(n)]
        lines = ["IndividualID " + " ".join(cols)]
        if gene:
            ref = rng.integers(0, 4, n)
            nfam = 6
            fams = [np.where(rng.random(n) < 0.08, rng.integers(0, 4, n), ref) for _ in range(nfam)]
            for ti, t in enumerate(types[gene]):
                seq = ["ACGT"[x] for x in fams[ti % nfam]]
                if ti % 7 == 3:
                    seq[5] = "_"
                if ti % 11 == 5:
                    seq[-3:] = ["*"] * 3
                lines.append(t + " " + " ".join(seq))
                content[(name, t)] = "".join(seq)
        else:
            lines.append("ref " + " ".join("ACGT"[x] for x in rng.integers(0, 4, n)))
        (prg / name).write_text("\n".join(lines) + "\n")
        base[name] = level
        level += n
    (prg / "notes.txt").write_text("not a segment\n")
    return segs, types, content, base, level


def expected_locus(segs, types, content, base, exon_files):
    """clusters as hla/HLATyper.cpp:1322-1372 builds them: alleles in name order, a new cluster per new combined sequence."""
    seq_of = {}
    for t in types:
        if ":" not in t:
            continue
        seq_of[t] = "".join(content[(f, t)] for f in exon_files)
    cluster_of_seq = {}; members = []; seqs = []
    for t in sorted(seq_of):
        s = seq_of[t]
        if s not in cluster_of_seq:
            cluster_of_seq[s] = len(members); members.append([]); seqs.append(s)
        members[cluster_of_seq[s]].append(t)
    return seq_of, [sorted(m) for m in members], seqs


def test_typer_levels_genes_and_locus(pkg, tmp_path):
    rng = np.random.default_rng(5)
    segs, types, content, base, n_levels = make_graph_dir(tmp_path, rng)
    lib = C.CDLL(pkg.LIB_PATH)
    T = pkg.Typer(lib, tmp_path)
    names = T.level_names()
    assert len(names) == n_levels and names[0] == "L0_1" and names[30] == "L30_2" and T.level_of("L31_2") == 31 and T.level_of("nope") == -1
    # gene level ranges in name order; "HLA-A" spans intron_1 .. exon_3
    assert T.genes() == [("DQA1", base["7_gene_DQA1_1_exon_2.txt"], base["7_gene_DQA1_1_exon_2.txt"] + 49), ("HLA-A", 30, 30 + 25 + 60 + 20 + 45 - 1)]
    L = T.locus("A")
    files = ["3_gene_HLA-A_2_exon_2.txt", "5_gene_HLA-A_4_exon_3.txt"]
    seq_of, members, seqs = expected_locus(segs, types["A"], content, base, files)
    assert L.n_types == len(seq_of) and L.n_clusters == len(members) and L.n_columns == 105 and L.n_exons == 2 and L.exon_length.tolist() == [60, 45]
    assert [L.cluster_id(c) for c in range(L.n_clusters)] == [";".join(m) for m in members]
    assert ["".join(map(chr, row)) for row in L.cluster_seq] == seqs
    assert L.type_cluster(members[2][0]) == 2 and L.type_cluster("A_odd_name") == -1
    lv = list(range(55, 115)) + list(range(135, 180))
    assert L.col_level.tolist() == lv and L.level_min == 55 and L.level_max == 179
    assert L.col_exon.tolist() == [0] * 60 + [1] * 45 and L.col_exon_pos.tolist() == list(range(60)) + list(range(45))
    l2e = np.full(125, -1); l2e[np.array(lv) - 55] = np.arange(105)
    assert np.array_equal(L.level_to_exon, l2e)
    # one-exon locus from the built-in table, explicit exon list, and failures
    D = T.locus("DQA1"); assert D.n_columns == 50 and D.n_types == 12
    A2 = T.locus("A", exons=["exon_3"]); assert A2.n_columns == 45 and A2.level_min == 135
    for bad in (("XYZ", None), ("A", ["exon_9"]), ("A", ["intron_1"])):
        with pytest.raises(pkg.HlalaError):
            T.locus(bad[0], exons=bad[1])
    with pytest.raises(pkg.HlalaError):
        pkg.Typer(lib, tmp_path / "missing")
    # k-mers of a cluster: exon by exon, gaps removed, the ones with '*' counted but not asked
    for c in (0, 3, L.n_clusters - 1):
        s = seqs[c]; want = []; total = 0
        for ex in (s[:60], s[60:]):
            ex = ex.replace("_", "")
            for i in range(len(ex) - 31 + 1):
                total += 1
                if "*" not in ex[i:i + 31]:
                    want.append(ex[i:i + 31])
        got, nt = L.cluster_kmers(c, 31)
        assert got == want and nt == total
    T.close()


def test_g_group_translation_in_bestguess(pkg, tmp_path):
    rng = np.random.default_rng(6)
    segs, types, content, base, n_levels = make_graph_dir(tmp_path, rng, n_types=8)
    lib = C.CDLL(pkg.LIB_PATH)
    T = pkg.Typer(lib, tmp_path)
    gfile = tmp_path / "hla_nom_g.txt"
    gfile.write_text("# file: hla_nom_g.txt\nA*;01:01:01/01:02/01:03;01:01:01G\nA*;01:04;\nA*;02:01:01/02:02;02:01:01G\nB*;07:02;\n")
    T.load_g_groups(gfile)
    L = T.locus("A")
    # run the writer with a trivial table to look at the translated names only
    Cn = L.n_clusters; nP = Cn * (Cn + 1) // 2
    e = synth_positions(rng, 30, L.n_columns); e["read_reverse"] = np.zeros(60, np.uint8); e["read_mapq"] = np.ones(60)
    ll = -rng.random(nP) * 50; ma = rng.random(nP); mm = np.floor(rng.random(nP) * 5)
    call = ob.call_locus(ll, ma, mm)
    co = pkg.CallOut(call["first_cluster"], call["second_cluster"], call["first_marginal"], call["second_p"], call["ll_max"], call["max_pair"], call["n_sort_ties"])
    out = tmp_path / "out"
    pkg.typer_begin_output(lib, out)
    L.write_files(out, e, ["r%d" % i for i in range(30)], ["r%d" % i for i in range(30)], ll, ma, mm, call["order"], call["p_normalized"], co)
    rows = (out / "R1_bestguess_G.txt").read_text().splitlines()
    assert rows[0].split("\t")[-1] == "perfectG" and len(rows) == 3
    amap = {"A*01:01:01": "A*01:01:01G", "A*01:02": "A*01:01:01G", "A*01:03": "A*01:01:01G", "A*01:04": "A*01:04", "A*02:01:01": "A*02:01:01G", "A*02:02": "A*02:01:01G"}
    for row, cl in zip(rows[1:], (call["first_cluster"], call["second_cluster"])):
        f = row.split("\t"); mem = L.cluster_id(cl).split(";")
        groups = {}
        for a in mem:
            if a in amap:
                groups[amap[a]] = groups.get(amap[a], 0) + 1
        if not groups:
            assert f[2] == ";".join(mem) and f[-1] == "0"
        elif len(groups) == 1:
            assert f[2] == list(groups)[0] and f[-1] == "1"
        else:
            assert f[2] in groups and groups[f[2]] == max(groups.values()) and f[-1] == "0"
    # a locus the G file does not know gets no G rows
    D = T.locus("DQA1"); Cn = D.n_clusters; nP = Cn * (Cn + 1) // 2
    e = synth_positions(rng, 20, D.n_columns); e["read_reverse"] = np.zeros(40, np.uint8); e["read_mapq"] = np.ones(40)
    ll = -rng.random(nP) * 50; ma = rng.random(nP); mm = np.floor(rng.random(nP) * 5); call = ob.call_locus(ll, ma, mm)
    co = pkg.CallOut(call["first_cluster"], call["second_cluster"], call["first_marginal"], call["second_p"], call["ll_max"], call["max_pair"], call["n_sort_ties"])
    D.write_files(out, e, ["r%d" % i for i in range(20)], None, ll, ma, mm, call["order"], call["p_normalized"], co)
    assert len((out / "R1_bestguess_G.txt").read_text().splitlines()) == 3 and len((out / "R1_bestguess.txt").read_text().splitlines()) == 5
    pkg.typer_end_output(lib, out, ["A", "DQA1"])
    assert (out / "R1_parameters.txt").read_text() == "Loci = A,DQA1\nveryConservativeReadLikelihoods = 0\n"


def expected_files(L, e, names1, names2, prm, use, tallies_src, ll, ma, mm, call, kmers, long_mode, un_cov=30, un_frac=0.2):
    """The writers of HLATypeInference as the reference has them: maps of maps of lists."""
    def geno(j):
        return bytes(e["geno_chars"][e["geno_off"][j]:e["geno_off"][j + 1]]).decode()
    pc = lambda q: -1 if q == 0 else 1 - math.exp(math.log(10.0) * ((q - 33) / -10.0))
    P = L.n_columns
    pile = {}
    for r in range(e["n_reads"]):
        for j in range(e["pos_off"][r], e["pos_off"][r + 1]):
            if not use[j] or (long_mode and e["pos_novel_gap"][j] >= 2):
                continue
            col = int(e["pos_exon"][j])
            pile.setdefault(int(L.col_exon[col]), {}).setdefault(int(L.col_exon_pos[col]), []).append((j, r))
    lines = []; utilized = set()
    for exon in sorted(pile):
        for xp in range(int(L.exon_length[exon])):
            if xp not in pile[exon]:
                lines.append("%d\t%d\t0" % (exon, xp)); continue
            ent = pile[exon][xp]; strs = []; counts = {}
            for j, r in ent:
                m = 1 if e["pos_mate"][j] == 2 else 0
                gt = geno(j)
                quals = [str(int(np.int8(q))) for q, ch in zip(e["qual_chars"][e["geno_off"][j]:e["geno_off"][j + 1]], gt) if ch != "_"]
                unit = int(e["read_pair"][r])
                this = (names2 if m else names1)[unit] if names2 is not None else names1[unit]
                other = ((names1 if m else names2)[unit]) if names2 is not None else ""
                strs.append(gt + " (" + ", ".join(quals) + ") [pairsDistance " + g(float(e["read_distance"][r])) + " | alignmentLength " + str(int(e["read_cols_nongap"][2 * r + m])) + " | " +
                            g(pc(int(e["pos_mapq"][j]))) + " | " + g(e["read_mapq"][2 * r + m]) + " " + g(e["read_mapq"][2 * r + m]) + " | " +
                            g(e["read_weighted_ok"][2 * r + m]) + " " + g(e["read_weighted_ok"][2 * r + 1 - m]) + " | " + this + " " + other + "]")
                utilized.add(this); counts.setdefault(gt, []).append(int(e["read_cols_nongap"][2 * r + m]))
            col = int(e["pos_exon"][ent[0][0]])
            summ = ""
            for a in sorted(counts):
                t = tallies_src[col][a]
                summ += a + "x" + str(len(counts[a])) + "[" + g(sum(counts[a]) / len(counts[a])) + ";" + g(min(t[1], t[0] - t[1]) / t[0]) + ";" + g(t[2] / t[0]) + "]"
            lines.append("%d\t%d\t%d\t%s\t%s" % (exon, xp, len(ent), ", ".join(strs), summ))
    files = {"R1_pileup_%s.txt" % L.name: "".join(l + "\n" for l in lines), "R1_readIDs_%s.txt" % L.name: "".join(u + "\n" for u in sorted(utilized))}
    Cn = L.n_clusters
    pairs = [(a, b) for a in range(Cn) for b in range(a, Cn)]
    ids = [L.cluster_id(c) for c in range(Cn)]
    files["R1_PP_%s_pairs.txt" % L.name] = "ClusterID\tP\tLL\tMismatches_avg\n" + "".join(
        "%s/%s\t%s\t%s\t%s\n" % (ids[pairs[i][0]], ids[pairs[i][1]], g(call["p_normalized"][i]), g(ll[i]), g(ma[i])) for i in call["order"])
    first, second = call["first_cluster"], call["second_cluster"]
    s1, s2 = L.cluster_seq[first], L.cluster_seq[second]
    cov = []; tot = []; bad = []; unacc = 0
    for col in range(P):
        ent = pile.get(int(L.col_exon[col]), {}).get(int(L.col_exon_pos[col]), [])
        a1, a2 = chr(s1[col]), chr(s2[col])
        cov.append(len(ent)); tot.append(len(ent)); bad.append(sum(1 for j, r in ent if geno(j) not in (a1, a2)))
        post = {a: t[3] for a, t in tallies_src.get(col, {}).items() if t[3] >= 0}
        if post and sum(post.values()) >= un_cov:
            for a, n in post.items():
                if a not in (a1, a2) and n / sum(post.values()) >= un_frac:
                    unacc += 1
    avg = sum(bad) / sum(tot) if sum(tot) else 0
    ce = "Column\tCoverage\tExpectedIncompatible\tObservedIncompatible\tp\n"
    for col in range(P):
        exp = avg * tot[col]; p = 1
        if bad[col] > exp:
            stat = ((tot[col] - bad[col]) - (tot[col] - exp)) ** 2 / (tot[col] - exp) + (bad[col] - exp) ** 2 / exp
            p = 1 - (1 - math.erfc(math.sqrt(stat / 2)))
        ce += "%d\t%d\t%s\t%d\t%s\n" % (col, tot[col], g(exp), bad[col], g(p))
    files["R1_columnIncompatibilities_%s.txt" % L.name] = ce
    sc = sorted(cov); bases_used = int(np.sum(use))
    a, b = min(first, second), max(first, second)
    q2 = -1 * mm[a * Cn - a * (a - 1) // 2 + (b - a)]
    common = "\t".join([g(bases_used / P), g(sc[int(len(sc) / 10.0)]), g(sc[0])])
    rows = ["\t".join([L.name, "1", ids[first], g(call["first_marginal"]), g(q2), common, g(kmers[0]), g(avg), str(unacc)]),
            "\t".join([L.name, "2", ids[second], g(call["second_p"]), g(q2), common, g(kmers[1]), g(avg), str(unacc)])]
    return files, rows


def tallies_of(e, prm, oracle_use_before_hc, long_mode):
    """per column: allele -> [count, reverse count, from-first count, post-filtering count or -1] at the stage before the high-coverage filter:
    obtained by running the oracle's filter with the high-coverage and strand filters switched off (same ignore sets as that stage)."""
    t = {}
    for r in range(e["n_reads"]):
        for j in range(e["pos_off"][r], e["pos_off"][r + 1]):
            if not oracle_use_before_hc[j]:
                continue
            col = int(e["pos_exon"][j]); a = bytes(e["geno_chars"][e["geno_off"][j]:e["geno_off"][j + 1]]).decode()
            x = t.setdefault(col, {}).setdefault(a, [0, 0, 0, -1])
            x[0] += 1
            m = 1 if e["pos_mate"][j] == 2 else 0
            x[1] += int(e["read_reverse"][2 * r + m]); x[2] += int(e["pos_mate"][j] == 1)
    for col, al in t.items():
        n = sum(x[0] for x in al.values())
        if n >= prm.high_coverage_min_coverage:
            for a, x in al.items():
                if not (prm.high_coverage_filter and x[0] / n < prm.high_coverage_min_freq):
                    x[3] = x[0]
    return t


@pytest.mark.parametrize("seed,n_reads,long_mode,hc", [(21, 400, False, False), (22, 3000, False, True), (23, 1500, True, False)])
def test_locus_files_match_reference_rules(pkg, oracle, tmp_path, seed, n_reads, long_mode, hc, monkeypatch):
    if seed != 21:
        monkeypatch.setenv("HLALA_PILEUP_RUN", "700")                     # the pile-up lines in several runs of columns formatted side by side, as for a real locus
    rng = np.random.default_rng(seed)
    segs, types, content, base, n_levels = make_graph_dir(tmp_path, rng)
    lib = C.CDLL(pkg.LIB_PATH)
    T = pkg.Typer(lib, tmp_path); L = T.locus("A")
    e = synth_positions(rng, n_reads, L.n_columns)
    e["read_reverse"] = (rng.random(2 * n_reads) < 0.4).astype(np.uint8)
    e["read_mapq"] = np.round(rng.random(2 * n_reads), 3)
    e["read_distance"] = rng.integers(-50, 400, n_reads).astype(np.int32)
    e["read_cols_nongap"] = rng.integers(100, 151, 2 * n_reads).astype(np.int32)
    e["pos_novel_gap"] = (rng.random(e["n_pos"]) < 0.05).astype(np.int32) * 2
    if long_mode:
        e["read_weighted_ok"][1::2] = -1; e["read_mapq"][1::2] = -1; e["pos_mate"][:] = 1
    kw = dict(high_coverage_filter=1, high_coverage_min_coverage=25, high_coverage_min_freq=0.1) if hc else dict(high_coverage_min_coverage=40)
    if long_mode:
        kw.update(long_read_strand_filter=1, strand_min_allele_coverage=20, strand_min_freq=0.3)
    prm = pkg.default_filter_params(**kw)
    use, ign, st = ob.filter_positions(e, prm)
    stage = pkg.default_filter_params(high_coverage_min_coverage=kw.get("high_coverage_min_coverage", 100))
    use_before, _, _ = ob.filter_positions(e, stage)                      # first-20 filter only: the stage the per-allele counts are taken at
    tal = tallies_of(e, prm, use_before, long_mode)
    Cn = L.n_clusters; nP = Cn * (Cn + 1) // 2
    ll = -np.round(rng.random(nP) * 300, 2); ma = np.round(rng.random(nP) * 3, 3); mm = np.floor(rng.random(nP) * 4)
    ll[rng.integers(0, nP, 5)] = ll.max()                                  # ties at the top
    call = ob.call_locus(ll, ma, mm)
    co = pkg.CallOut(call["first_cluster"], call["second_cluster"], call["first_marginal"], call["second_p"], call["ll_max"], call["max_pair"], call["n_sort_ties"])
    names1 = ["read%05d/1" % i for i in range(n_reads)]; names2 = None if long_mode else ["read%05d/2" % i for i in range(n_reads)]
    kmers = (0.9875, -1.0)
    out = tmp_path / "out"
    pkg.typer_begin_output(lib, out)
    res = L.write_files(out, e, names1, names2, ll, ma, mm, call["order"], call["p_normalized"], co, kmers_covered=kmers, params=prm, long_read_mode=long_mode,
                        unaccounted_min_coverage=20, unaccounted_min_fraction=0.05)
    files, rows = expected_files(L, e, names1, names2, prm, use, tal, ll, ma, mm, call, kmers, long_mode, un_cov=20, un_frac=0.05)
    for fn, text in files.items():
        got = (out / fn).read_text()
        if got != text:
            gl, tl = got.splitlines(), text.splitlines()
            for i, (x, y) in enumerate(zip(gl, tl)):
                assert x == y, (fn, i)
            assert len(gl) == len(tl), fn
    bg = (out / "R1_bestguess.txt").read_text().splitlines()
    assert bg[0] == "Locus\tChromosome\tAllele\tQ1\tQ2\tAverageCoverage\tCoverageFirstDecile\tMinimumCoverage\tproportionkMersCovered\tLocusAvgColumnError\tNColumns_UnaccountedAllele_fGT0.2"
    assert bg[1:] == rows
    assert res.bases_used == st["bases_used"] and res.n_piled_positions > 100 and res.n_utilized_reads > 10
    if hc:
        assert res.n_columns_unaccounted > 0


def canonical(kmer):
    rc = kmer[::-1].translate(str.maketrans("ACGTN", "TGCAN"))
    return min(kmer, rc)


@pytest.mark.gpu
def test_kmer_presence_matches_a_hash_of_all_read_kmers(pkg):
    """proportionkMersCovered (hla/HLATyper.cpp:999-1027, 2652-2688): the answers of the index the reference builds, without building it."""
    from tools import synth
    rng = np.random.default_rng(8)
    w = synth.make_world(seed=3, G=3000, k=1)
    b = synth.make_batch(w, 300, seed=9)
    b["read_bases"] = b["read_bases"].copy(); b["read_bases"][rng.integers(0, len(b["read_bases"]), 200)] = ord("N")
    ctx = pkg.Context(w["graph"], w["contigs"], insert_mean=b["insert_mean"], insert_sd=b["insert_sd"])
    gb = ctx.batch(b)
    for k in (31, 21, 5):
        for mask in (None, (np.arange(300) % 2).astype(np.uint8)):
            index = set()
            for r in range(600):
                if mask is not None and not mask[r // 2]:
                    continue
                s = bytes(b["read_bases"][b["read_off"][r]:b["read_off"][r + 1]]).decode()
                for i in range(len(s) - k + 1):
                    index.add(canonical(s[i:i + k]))
            reads = [bytes(b["read_bases"][b["read_off"][r]:b["read_off"][r + 1]]).decode() for r in rng.integers(0, 600, 150)]
            q = []
            for s in reads:
                i = int(rng.integers(0, len(s) - k + 1)); km = s[i:i + k]
                q.append(km if rng.random() < 0.5 else km[::-1].translate(str.maketrans("ACGTN", "TGCAN")))
            q += ["".join(rng.choice(list("ACGT"), k)) for _ in range(150)]
            q += ["A" * (k - 1) + "*", "N" * k]
            q += q[:10]                                                     # repeated questions
            got = ctx.kmer_presence(gb, q, k, mask)
            want = np.array([1 if (set(x) <= set("ACGT") and canonical(x) in index) else 0 for x in q], np.uint8)
            assert np.array_equal(got, want), (k, mask is None)
            assert want.sum() > 50 and (k < 10 or (want == 0).sum() > 50)
    with pytest.raises(pkg.HlalaError):
        ctx.kmer_presence(gb, ["A" * 32], 32)


@pytest.mark.gpu
def test_kmer_presence_of_kept_reads_is_the_union_over_the_batches(pkg):
    """hlala_kmer_keep_reads / hlala_kmer_presence_kept: the questions of HLATyper.cpp:2652-2688 asked after the batches were released --
    equal to the OR of hlala_kmer_presence over the batches (paired batches with and without a mask, an unpaired one, an empty selection)."""
    from tools import synth
    rng = np.random.default_rng(18)
    w = synth.make_world(seed=3, G=3000, k=1)
    ctx = pkg.Context(w["graph"], w["contigs"], insert_mean=200.0, insert_sd=35.0)
    assert np.array_equal(ctx.kmer_presence_kept(["A" * 31, "C" * 31]), [0, 0])                   # nothing kept yet
    batches, masks = [], []
    for i, n in enumerate((300, 170, 64)):
        b = synth.make_batch(w, n, seed=20 + i)
        b["read_bases"] = b["read_bases"].copy(); b["read_bases"][rng.integers(0, len(b["read_bases"]), 50)] = ord("N")
        batches.append(b); masks.append([(np.arange(n) % 3 != 0).astype(np.uint8), None, np.zeros(n, np.uint8)][i])
    for k in (31, 12):
        ctx.kmer_forget_reads()
        q = []
        for b in batches:
            for r in rng.integers(0, len(b["read_off"]) - 1, 60):
                s = bytes(b["read_bases"][b["read_off"][r]:b["read_off"][r + 1]]).decode()
                i = int(rng.integers(0, len(s) - k + 1)); q.append(s[i:i + k])
        q += ["".join(rng.choice(list("ACGT"), k)) for _ in range(100)] + ["N" * k]
        want = np.zeros(len(q), np.uint8); kept = []
        for b, m in zip(batches, masks):
            gb = ctx.batch(b)
            want |= ctx.kmer_presence(gb, q, k, m)
            kept.append(ctx.kmer_keep_reads(gb, m))
            del gb                                                          # the batch is gone when the questions are asked
        assert kept == [2 * int(masks[0].sum()), 2 * 170, 0]
        got = ctx.kmer_presence_kept(q, k)
        assert np.array_equal(got, want), k
        assert 50 < want.sum() < len(q) - (50 if k > 20 else 0)
    ctx.kmer_forget_reads()
    assert ctx.kmer_presence_kept(q, 12).sum() == 0
    with pytest.raises(pkg.HlalaError):
        ctx.kmer_presence_kept(["A" * 32], 32)


def synth_unit_stats(rng, n):
    f = np.where(rng.random(2 * n) < 0.4, 1.0, np.round(1 - rng.random(2 * n) * 0.1, 4))
    return dict(valid=(rng.random(n) < 0.95).astype(np.uint8), strands_valid=(rng.random(n) < 0.9).astype(np.uint8), distance=rng.integers(-100, 600, n).astype(np.int32),
                fraction_ok=f, weighted_ok=np.round(1 - rng.random(2 * n) * 0.05, 5), n_columns=rng.integers(140, 2000, 2 * n).astype(np.int32), mate_mapq=np.round(rng.random(2 * n), 3))


@pytest.mark.parametrize("unpaired", [False, True])
def test_summary_statistics_file(pkg, tmp_path, unpaired):
    """summaryStatistics.txt against hla/HLATyper.cpp:1030-1125 written out in Python."""
    rng = np.random.default_rng(31); n = 500
    st = synth_unit_stats(rng, n); mask = (rng.random(n) < 0.8).astype(np.uint8)
    lib = C.CDLL(pkg.LIB_PATH)
    out = tmp_path / "o"
    pkg.typer_write_summary(lib, out, st, unpaired=unpaired, unit_mask=mask, insert_mean=210.5, insert_sd=30.25)
    units = [u for u in range(n) if mask[u] and st["valid"][u]]
    if not unpaired:
        sv = [u for u in units if st["strands_valid"][u]]
        dist = sorted(float(st["distance"][u]) for u in sv)
        ok = sum(1 for u in sv if abs(float(st["distance"][u]) - 210.5) <= 5 * 30.25)
        S = 0.0
        for u in units:
            S += st["fraction_ok"][2 * u]; S += st["fraction_ok"][2 * u + 1]
        perfect = sum(int(st["fraction_ok"][2 * u] == 1) + int(st["fraction_ok"][2 * u + 1] == 1) for u in units)
        one = sum(1 for u in units if st["fraction_ok"][2 * u] == 1 or st["fraction_ok"][2 * u + 1] == 1)
        s = 0.0
        for d in dist:
            s += d
        exp = ("\nRead alignment statistics:\n\t - Total number (paired) alignments:                 %d\n" % len(units) +
               "\t\t - Alignment pairs with strands OK:                  %d (%s%%)\n" % (len(sv), g(len(sv) / len(units) * 100)) +
               "\t\t - Alignment pairs with strands OK && distance OK:   %d (%s%%)\n" % (ok, g(ok / len(units) * 100)) +
               "\t\t - Alignment pairs with strands OK, mean distance:   %s\n" % g(s / len(dist)) +
               "\t\t - Alignment pairs with strands OK, median distance: %s\n" % g(dist[len(dist) // 2]) +
               "\t\t - Alignment pairs, average fraction alignment OK:   %s\n" % g(S / (2.0 * len(units))) +
               "\t\t - Alignment pairs, at least one alignment perfect:   %d\n" % one +
               "\t\t - Single alignments, perfect (total):   %d (%d)\n" % (perfect, 2 * len(units)) +
               "\t - Total number (unpaired) alignments:                 0\n\t\t - Alignment pairs, average fraction alignment OK:   0\n"
               "\t\t - Single alignments, perfect (total):   0 (0)\n\t\t - Alignments with length >= 1000:   0\n")
    else:
        S = 0.0
        for u in units:
            S += st["fraction_ok"][2 * u]
        exp = ("\nRead alignment statistics:\n\t - Total number (paired) alignments:                 0\n"
               "\t\t - Alignment pairs with strands OK:                  0 (%s%%)\n\t\t - Alignment pairs with strands OK && distance OK:   0 (%s%%)\n" % ("-nan", "-nan") +
               "\t\t - Alignment pairs with strands OK, mean distance:   0\n\t\t - Alignment pairs with strands OK, median distance: 0\n"
               "\t\t - Alignment pairs, average fraction alignment OK:   0\n\t\t - Alignment pairs, at least one alignment perfect:   0\n"
               "\t\t - Single alignments, perfect (total):   0 (0)\n" +
               "\t - Total number (unpaired) alignments:                 %d\n" % len(units) +
               "\t\t - Alignment pairs, average fraction alignment OK:   %s\n" % g(S / len(units)) +
               "\t\t - Single alignments, perfect (total):   %d (%d)\n" % (sum(1 for u in units if st["fraction_ok"][2 * u] == 1), 2 * len(units)) +
               "\t\t - Alignments with length >= 1000:   %d\n" % sum(1 for u in units if st["n_columns"][2 * u] >= 1000))
    got = (out / "summaryStatistics.txt").read_text()
    assert got.replace("(nan%)", "(-nan%)") == exp          # 0/0 * 100 prints as the C library prints it (sign of the NaN is platform detail)


def test_histogram_lines(pkg, oracle, tmp_path):
    """histogram_matchesPerRead.txt: "read" / "readPair" lines of the pairs passing the pair test of the locus, then a "base" line per piled position."""
    rng = np.random.default_rng(41)
    make_graph_dir(tmp_path, rng)
    lib = C.CDLL(pkg.LIB_PATH)
    T = pkg.Typer(lib, tmp_path); L = T.locus("A")
    n_units = 900; n_reads = 300
    e = synth_positions(rng, n_reads, L.n_columns)
    e["read_pair"] = np.sort(rng.choice(n_units, n_reads, replace=False)).astype(np.int32)
    e["read_reverse"] = (rng.random(2 * n_reads) < 0.4).astype(np.uint8); e["read_mapq"] = np.round(rng.random(2 * n_reads), 3)
    st = synth_unit_stats(rng, n_units); mask = (rng.random(n_units) < 0.7).astype(np.uint8)
    prm = pkg.default_filter_params()
    use, ign, fs = ob.filter_positions(e, prm)
    Cn = L.n_clusters; nP = Cn * (Cn + 1) // 2
    ll = -np.round(rng.random(nP) * 300, 2); ma = np.round(rng.random(nP) * 3, 3); mm = np.floor(rng.random(nP) * 4)
    call = ob.call_locus(ll, ma, mm)
    co = pkg.CallOut(call["first_cluster"], call["second_cluster"], call["first_marginal"], call["second_p"], call["ll_max"], call["max_pair"], call["n_sort_ties"])
    names = ["u%04d" % i for i in range(n_units)]
    out = tmp_path / "out"; pkg.typer_begin_output(lib, out)
    kw = dict(insert_mean=250.0, insert_sd=40.0, min_mapq=0.2, min_weighted_ok=0.96)
    L.write_files(out, e, names, names, ll, ma, mm, call["order"], call["p_normalized"], co, params=prm, unit_stats=st, unit_mask=mask, **kw)
    exp = ["Locus\tLevelValue"]
    for u in range(n_units):
        if not mask[u] or not st["valid"][u]:
            continue
        w1, w2 = st["weighted_ok"][2 * u], st["weighted_ok"][2 * u + 1]
        if st["strands_valid"][u] and abs(float(st["distance"][u]) - 250.0) <= 5 * 40.0 and st["mate_mapq"][2 * u] >= 0.2 and w1 >= 0.96 and w2 >= 0.96:
            exp += ["A\tread" + g(w1), "A\tread" + g(w2), "A\treadPair" + g((w1 + w2) / 2.0)]
    for r in range(n_reads):
        for j in range(e["pos_off"][r], e["pos_off"][r + 1]):
            if use[j]:
                exp.append("A\tbase" + g(e["read_weighted_ok"][2 * r + (1 if e["pos_mate"][j] == 2 else 0)]))
    got = (out / "histogram_matchesPerRead.txt").read_text().splitlines()
    assert got == exp and len(exp) > 1000
