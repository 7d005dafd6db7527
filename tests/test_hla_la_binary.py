"""The `HLA-LA` host program (hla-la_amd/host/HLA-LA.cpp): the process-level contract HLA-LA.pl relies on.

CPU: `--action testBinary` prints exactly what the installation check expects (HLA-LA.cpp:131, README.md:76-78); a missing action or
missing --bwa_bin ends with a non-zero status (HLA-LA.pl:567-570 treats that as failure); `--action prepareGraph` leaves
<graph dir>/serializedGRAPH (HLA-LA.pl:254-257 checks its existence) holding exactly the arrays of PRG/graph.txt; the G-group table of the
reference (hla_nom_g.txt, 15 695 lines of data kept as a fixture) translates hand-checked alleles.
GPU: `--action HLA` with stand-ins for bwa and samtools that hand over a prepared BAM: the files under hla/ equal, byte for byte, the files
the ctypes path writes from the same inputs; reads_per_level.txt and the `Speed:` line are there; a run in three GPU batches writes the
same files as the run in one batch (buffer-pool reuse, per-batch chain numbering of the random seeds).
"""
import ctypes as C
import os
import stat
import subprocess

import numpy as np
import pytest

from tools import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "hla-la_amd", "bin", "HLA-LA")


@pytest.fixture(scope="module")
def exe():
    if not os.path.exists(EXE):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "hla-la_amd", "csrc"), "../bin/HLA-LA"])
    return EXE


def test_test_binary_action(exe):
    r = subprocess.run([exe, "--action", "testBinary"], capture_output=True)
    assert r.returncode == 0 and r.stdout == b"\nHLA*LA binary functional!\n\n"
    # unknown argument names are ignored (HLA-LA.cpp:71-79)
    r = subprocess.run([exe, "--somethingElse", "1", "--action", "testBinary"], capture_output=True)
    assert r.returncode == 0 and r.stdout == b"\nHLA*LA binary functional!\n\n"


def test_errors_end_with_a_non_zero_status(exe, tmp_path):
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode != 0 and "Missing --action parameter" in r.stderr
    r = subprocess.run([exe, "--action", "HLA", "--sampleID", "s", "--outputDirectory", str(tmp_path / "o"), "--PRG_graph_dir", str(tmp_path)], capture_output=True, text=True)
    assert r.returncode != 0 and "Please specify arguments --bwa_bin" in r.stderr
    r = subprocess.run([exe, "--action", "HLA", "--bwa_bin", "/bin/true", "--sampleID", "s", "--outputDirectory", str(tmp_path / "o"), "--PRG_graph_dir", str(tmp_path)],
                       capture_output=True, text=True)
    assert r.returncode != 0 and "Please specify arguments --samtools_bin" in r.stderr
    r = subprocess.run([exe, "--action", "prepareGraph", "--PRG_graph_dir", str(tmp_path / "nowhere")], capture_output=True, text=True)
    assert r.returncode != 0 and "graph.txt" in r.stderr
    r = subprocess.run([exe, "--action", "KIR", "--bwa_bin", "/bin/true", "--samtools_bin", "/bin/true"], capture_output=True, text=True)
    assert r.returncode != 0 and "not part of this build" in r.stderr
    r = subprocess.run([exe, "--action"], capture_output=True, text=True)
    assert r.returncode != 0


def test_prepare_graph_leaves_serialized_graph(exe, pkg, tmp_path):
    from test_graph_files import write_graph_txt
    w = synth.make_world(seed=3, G=2500, k=2)
    (tmp_path / "PRG").mkdir()
    write_graph_txt(tmp_path / "PRG" / "graph.txt", w["graph"], np.random.default_rng(2))
    r = subprocess.run([exe, "--action", "prepareGraph", "--PRG_graph_dir", str(tmp_path)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert (tmp_path / "serializedGRAPH").exists() and (tmp_path / "serializedGRAPH_preGapPathIndex").exists()
    lib = C.CDLL(pkg.LIB_PATH)
    a = pkg.load_graph_text(lib, tmp_path / "PRG" / "graph.txt"); b = pkg.load_graph_cache(lib, tmp_path / "serializedGRAPH")
    assert a["n_levels"] == b["n_levels"]
    for k in ("node_level", "edge_from", "edge_to", "edge_label"):
        assert np.array_equal(a[k], b[k]), k


def test_g_groups_of_the_reference_file(pkg, tmp_path):
    """hla_nom_g.txt as the reference ships it (IPD-IMGT/HLA 3.32.0): translations checked by hand against the file's lines."""
    from test_typer_files import make_graph_dir
    make_graph_dir(tmp_path, np.random.default_rng(6), n_types=4)
    lib = C.CDLL(pkg.LIB_PATH)
    T = pkg.Typer(lib, tmp_path)
    with pytest.raises(pkg.HlalaError):
        T.g_translate(["A*01:01:01:01"])                                   # no table loaded yet
    T.load_g_groups(os.path.join(ROOT, "tests", "golden", "hla_nom_g.txt"))
    # line 7: A*;01:01:01:01/01:01:01:02N/.../01:253;01:01:01G
    assert T.g_translate(["A*01:01:01:01"]) == ("A*01:01:01G", True)
    assert T.g_translate(["A*01:01:01:02N", "A*01:04N", "A*01:253"]) == ("A*01:01:01G", True)
    # line 8: A*;01:01:02;  -- an allele without a group is its own code
    assert T.g_translate(["A*01:01:02"]) == ("A*01:01:02", True)
    # two groups: the more frequent one, not perfect
    assert T.g_translate(["A*01:01:01:01", "A*01:01:51", "A*01:01:02"]) == ("A*01:01:01G", False)
    assert T.g_translate(["A*02:01:01:01"]) == ("A*02:01:01G", True) and T.g_translate(["B*07:02:01:01", "B*07:44N"]) == ("B*07:02:01G", True)
    assert T.g_translate(["C*07:02:01:01"]) == ("C*07:02:01G", True) and T.g_translate(["DRB1*15:01:01:01", "DRB1*15:146"]) == ("DRB1*15:01:01G", True)
    assert T.g_translate(["DQA1*01:01:02", "DQA1*01:12"]) == ("DQA1*01:01:01G", True)
    # nothing known: the list itself; a locus the file does not have: refused like can_translateToG_locus
    assert T.g_translate(["A*99:99", "A*98:98"]) == ("A*99:99;A*98:98", False)
    with pytest.raises(pkg.HlalaError):
        T.g_translate(["KIR2DL1*001"])
    with pytest.raises(pkg.HlalaError):
        T.g_translate(["H*01:01"])                                         # HLA-H is typed by the reference but has no G groups
    T.close()


# ------------------------------------------------------------------------------------------------------------ GPU

def _stub(path, text):
    path.write_text(text)
    path.chmod(path.stat().st_mode | stat.S_IXUSR | stat.S_IXGRP | stat.S_IXOTH)


@pytest.mark.gpu
def test_action_hla_writes_the_files_of_the_ctypes_path(exe, pkg, tmp_path):
    from test_bam import batch_records, write_bam
    from test_end_to_end import write_graph_dir
    from test_graph_files import write_contigs_dir, write_graph_txt
    gdir = tmp_path / "graph"; gdir.mkdir()
    G = 4000; exons = [(1200, 1470), (1900, 2176)]
    w = synth.make_world(seed=12, G=G, k=1, n_mut=6, mut_density=0.03)
    lib = C.CDLL(pkg.LIB_PATH)
    write_graph_dir(gdir, w["H"], exons)
    write_graph_txt(gdir / "PRG" / "graph.txt", w["graph"], np.random.default_rng(2))
    write_contigs_dir(gdir, w["contigs"], np.random.default_rng(3))
    b = synth.make_batch(w, 700, seed=77, haps=(2, 5))
    contigs, intervals = pkg.load_contigs_dir(lib, gdir, extended_reference_genome=False)
    clen = np.diff(w["contigs"]["contig_off"])
    bam = tmp_path / "premade.bam"
    write_bam(bam, [(iv[0], int(clen[i])) for i, iv in enumerate(intervals)], batch_records(b, np.random.default_rng(1)), block=30000)
    # ---- stand-ins for bwa and samtools: the command lines of BWAmapper::map run through them unchanged
    log = tmp_path / "cmds.log"
    _stub(tmp_path / "bwa", f"#!/bin/bash\necho \"bwa $@\" >> {log}\nif [ \"$1\" = index ]; then touch $2.sa $2.ann $2.bwt; fi\nexit 0\n")
    _stub(tmp_path / "samtools", f"#!/bin/bash\necho \"samtools $@\" >> {log}\ncase \"$1\" in\n view) cat > /dev/null ;;\n sort) while [ $# -gt 0 ]; do if [ \"$1\" = -o ]; then cp {bam} \"$2\"; fi; shift; done ;;\n"
                                  " index) touch \"$2.bai\" ;;\nesac\nexit 0\n")
    (tmp_path / "r1.fq").write_text("@r\nA\n+\nI\n"); (tmp_path / "r2.fq").write_text("@r\nA\n+\nI\n")
    out1 = tmp_path / "out1"
    base = [exe, "--action", "HLA", "--maxThreads", "2", "--sampleID", "S1", "--PRG_graph_dir", str(gdir), "--FASTQU", str(tmp_path / "r1.fq"), "--FASTQ1", str(tmp_path / "r1.fq"),
            "--FASTQ2", str(tmp_path / "r2.fq"), "--bwa_bin", str(tmp_path / "bwa"), "--samtools_bin", str(tmp_path / "samtools"), "--mapAgainstCompleteGenome", "0", "--longReads", "0",
            "--loci", "A", "--rngSeed", "5"]
    r = subprocess.run(base + ["--outputDirectory", str(out1)], capture_output=True, text=True, timeout=600, cwd=str(tmp_path))
    assert r.returncode == 0, r.stdout + r.stderr
    cmds = log.read_text().splitlines()
    ref = str(gdir / "mapping_PRGonly" / "referenceGenome.fa")
    assert cmds[0] == f"bwa index {ref}"
    # (the two sides of the pipe start concurrently: their log lines come in either order)
    assert sorted(cmds[1:3]) == sorted([f"bwa mem -t2 -M -a {ref} {tmp_path / 'r1.fq'} {tmp_path / 'r2.fq'}", "samtools view -@ 1 -Sb -"])
    assert cmds[3] == f"samtools sort -@ 2 -o {out1 / 'remapped_with_a.bam'} {out1 / 'remapped_with_a.bam.unsorted'}" and cmds[4] == f"samtools index {out1 / 'remapped_with_a.bam'}"
    assert "Speed: " in r.stdout and " protoSeeds (read pairs) per s" in r.stdout and "Processed 700 protoSeeds (read pairs) / 0 protoSeeds (unpaired long reads)" in r.stdout
    # ---- the same through the ctypes binding (the reference's default filter parameters, as the binary uses them)
    graph = pkg.load_graph_text(lib, gdir / "PRG" / "graph.txt")
    seeds, names, cnt = pkg.bam_extract_seeds(lib, bam, intervals)
    ctx0 = pkg.Context(graph, contigs, insert_mean=200.0, insert_sd=35.0, rng_seed=5)
    ins = ctx0.estimate_insert_size(seeds)
    ctx = pkg.Context(graph, contigs, insert_mean=ins["mean"], insert_sd=ins["sd"], rng_seed=5)
    gb = ctx.batch(seeds); gb.align()
    T = pkg.Typer(lib, gdir); L = T.locus("A")
    genes = T.genes(); ctx.set_gene_intervals([g[1] for g in genes], [g[2] for g in genes])
    include = gb.postprocess()
    e = gb.exon_positions(L.level_min, L.level_to_exon, ins["mean"], ins["sd"], pair_mask=include)
    prm = pkg.default_filter_params()
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
    assert files == sorted(os.listdir(out1 / "hla")) and "R1_bestguess.txt" in files and len(files) == 9
    for fn in files:
        assert (out_py / fn).read_bytes() == (out1 / "hla" / fn).read_bytes(), fn
    # reads_per_level.txt: level \t level name \t bases (processBAM.cpp:1902-1913), the counters of the context
    cov = ctx.coverage()
    rows = [l.split("\t") for l in (out1 / "reads_per_level.txt").read_text().splitlines()]
    assert len(rows) == G and [int(x[0]) for x in rows] == list(range(G)) and rows[5][1] == "L5" and np.array_equal(np.array([int(x[2]) for x in rows]), cov)
    assert (out1 / "remapped_with_a.bam").exists() and (out1 / "remapped_with_a.bam.bai").exists()
    # ---- three GPU batches through one context: the same files
    out3 = tmp_path / "out3"
    r3 = subprocess.run(base + ["--outputDirectory", str(out3), "--batchPairs", "300"], capture_output=True, text=True, timeout=600, cwd=str(tmp_path))
    assert r3.returncode == 0 and "in 3 GPU batch(es)" in r3.stdout, r3.stdout + r3.stderr
    for fn in files:
        assert (out3 / "hla" / fn).read_bytes() == (out1 / "hla" / fn).read_bytes(), fn
    assert (out3 / "reads_per_level.txt").read_bytes() == (out1 / "reads_per_level.txt").read_bytes()
    # ---- two contexts (listed devices 0,0: the multi-GPU walk of the host program, one host thread per context, batches dealt round-robin,
    # coverage summed, exon positions merged in batch order, call on the first context) write the same files as one context
    out2 = tmp_path / "out2"
    r2 = subprocess.run(base + ["--outputDirectory", str(out2), "--batchPairs", "150", "--devices", "0,0", "--decodeThreads", "3"], capture_output=True, text=True, timeout=600, cwd=str(tmp_path))
    assert r2.returncode == 0 and "in 5 GPU batch(es) on 2 device context(s)" in r2.stdout and "End-to-end: " in r2.stdout, r2.stdout + r2.stderr
    for fn in files:
        assert (out2 / "hla" / fn).read_bytes() == (out1 / "hla" / fn).read_bytes(), fn
    assert (out2 / "reads_per_level.txt").read_bytes() == (out1 / "reads_per_level.txt").read_bytes()
    # ---- four contexts (what `--devices 0,1,2,3` does on a node with four GPUs), seven batches dealt round-robin
    out4 = tmp_path / "out4"
    r24 = subprocess.run(base + ["--outputDirectory", str(out4), "--batchPairs", "100", "--devices", "0,0,0,0"], capture_output=True, text=True, timeout=600, cwd=str(tmp_path))
    assert r24.returncode == 0 and "on 4 device context(s)" in r24.stdout, r24.stdout + r24.stderr
    for fn in files:
        assert (out4 / "hla" / fn).read_bytes() == (out1 / "hla" / fn).read_bytes(), fn
    assert (out4 / "reads_per_level.txt").read_bytes() == (out1 / "reads_per_level.txt").read_bytes()
    # ---- BASELINE config 4 in small: two samples in one call, one per listed device, side by side; each writes what its own call writes
    ra = [x for x in base]
    for key, val in (("--sampleID", "S1,S2"), ("--FASTQ1", f"{tmp_path / 'r1.fq'},{tmp_path / 'r1.fq'}"), ("--FASTQ2", f"{tmp_path / 'r2.fq'},{tmp_path / 'r2.fq'}"),
                     ("--FASTQU", f"{tmp_path / 'r1.fq'},{tmp_path / 'r1.fq'}")):
        ra[ra.index(key) + 1] = val
    r4 = subprocess.run(ra + ["--outputDirectory", f"{tmp_path / 'outA'},{tmp_path / 'outB'}", "--devices", "0,0"], capture_output=True, text=True, timeout=600, cwd=str(tmp_path))
    assert r4.returncode == 0 and "Processed 2 samples on 2 device(s)" in r4.stdout, r4.stdout + r4.stderr
    assert "Graph directory read once for 2 samples" in r4.stdout                    # the samples share one view of the graph directory
    for o in ("outA", "outB"):
        for fn in files:
            assert (tmp_path / o / "hla" / fn).read_bytes() == (out1 / "hla" / fn).read_bytes(), (o, fn)
    # ---- a stale file under hla/ is wiped (processBAM.cpp:1805-1806), a failing mapper ends with a non-zero status
    (out1 / "hla" / "stale.txt").write_text("x")
    r = subprocess.run(base + ["--outputDirectory", str(out1)], capture_output=True, text=True, timeout=600, cwd=str(tmp_path))
    assert r.returncode == 0 and not (out1 / "hla" / "stale.txt").exists()
    # (a failing bwa inside `bwa mem | samtools view` goes unnoticed in the reference too: the status of a pipeline is its last command's)
    _stub(tmp_path / "samtools_fail", "#!/bin/bash\nif [ \"$1\" = sort ]; then exit 3; fi\ncat > /dev/null\nexit 0\n")
    bad = [x if x != str(tmp_path / "samtools") else str(tmp_path / "samtools_fail") for x in base]
    r = subprocess.run(bad + ["--outputDirectory", str(tmp_path / "outbad")], capture_output=True, text=True, timeout=600, cwd=str(tmp_path))
    assert r.returncode != 0 and "returned code" in r.stderr


@pytest.mark.gpu
def test_eight_samples_on_eight_contexts_in_one_call(exe, pkg, tmp_path):
    """BASELINE config 4 in its own shape on a one-GPU box: EIGHT DIFFERENT samples in one `HLA-LA --action HLA` call on `--devices 0,0,0,0,0,0,0,0` -- eight contexts
    (eight sets of DP slabs and pools) in one process, eight sample threads side by side, three GPU batches per sample, the graph directory read once -- and every
    sample's hla/* and reads_per_level.txt are the bytes of its own single-sample call.  The reference adds up the parts of its per-thread runs the same way
    (mapper/processBAM.cpp:1866-1887); on an eight-GPU node the only difference is the device numbers."""
    from test_bam import batch_records, write_bam
    from test_end_to_end import write_graph_dir
    from test_graph_files import write_contigs_dir, write_graph_txt
    gdir = tmp_path / "graph"; gdir.mkdir()
    G = 4000; exons = [(1200, 1470), (1900, 2176)]
    w = synth.make_world(seed=12, G=G, k=1, n_mut=6, mut_density=0.03)
    lib = C.CDLL(pkg.LIB_PATH)
    write_graph_dir(gdir, w["H"], exons)
    write_graph_txt(gdir / "PRG" / "graph.txt", w["graph"], np.random.default_rng(2))
    write_contigs_dir(gdir, w["contigs"], np.random.default_rng(3))
    contigs, intervals = pkg.load_contigs_dir(lib, gdir, extended_reference_genome=False)
    clen = np.diff(w["contigs"]["contig_off"])
    NS = 8
    names = [f"s{i}" for i in range(NS)]
    for i, nm in enumerate(names):       # eight samples drawn from different haplotype pairs, different sizes
        b = synth.make_batch(w, 220 + 20 * i, seed=500 + i, haps=(i % 5, (i + 2) % 5 + 1))
        write_bam(tmp_path / f"premade_{nm}.bam", [(iv[0], int(clen[k])) for k, iv in enumerate(intervals)], batch_records(b, np.random.default_rng(10 + i)), block=30000)
    # stand-ins for bwa and samtools: `samtools sort -o <outdir>/remapped_with_a.bam` receives the BAM made for the sample whose output directory it is
    _stub(tmp_path / "bwa", "#!/bin/bash\nif [ \"$1\" = index ]; then touch $2.sa $2.ann $2.bwt; fi\nexit 0\n")
    _stub(tmp_path / "samtools", f"#!/bin/bash\ncase \"$1\" in\n view) cat > /dev/null ;;\n sort) while [ $# -gt 0 ]; do if [ \"$1\" = -o ]; then d=$(basename $(dirname \"$2\")); cp {tmp_path}/premade_${{d#*_}}.bam \"$2\"; fi; shift; done ;;\n"
                                  " index) touch \"$2.bai\" ;;\nesac\nexit 0\n")
    (tmp_path / "r1.fq").write_text("@r\nA\n+\nI\n"); (tmp_path / "r2.fq").write_text("@r\nA\n+\nI\n")
    fq1 = str(tmp_path / "r1.fq"); fq2 = str(tmp_path / "r2.fq")

    def args(sample_ids, outs):
        n = len(sample_ids)
        return [exe, "--action", "HLA", "--maxThreads", "2", "--sampleID", ",".join(sample_ids), "--PRG_graph_dir", str(gdir), "--FASTQU", ",".join([fq1] * n), "--FASTQ1", ",".join([fq1] * n),
                "--FASTQ2", ",".join([fq2] * n), "--bwa_bin", str(tmp_path / "bwa"), "--samtools_bin", str(tmp_path / "samtools"), "--mapAgainstCompleteGenome", "0", "--longReads", "0",
                "--loci", "A", "--rngSeed", "5", "--batchPairs", "100", "--outputDirectory", ",".join(str(o) for o in outs)]
    outs8 = [tmp_path / f"all_{nm}" for nm in names]
    r8 = subprocess.run(args(names, outs8) + ["--devices", ",".join(["0"] * NS)], capture_output=True, text=True, timeout=900, cwd=str(tmp_path))
    assert r8.returncode == 0, r8.stdout[-3000:] + r8.stderr[-3000:]
    assert f"Processed {NS} samples on {NS} device(s)" in r8.stdout and f"Graph directory read once for {NS} samples" in r8.stdout
    seen = set()
    for nm, o8 in zip(names, outs8):
        o1 = tmp_path / f"one_{nm}"
        r1 = subprocess.run(args([nm], [o1]), capture_output=True, text=True, timeout=600, cwd=str(tmp_path))
        assert r1.returncode == 0 and ("in 3 GPU batch(es)" in r1.stdout or "in 4 GPU batch(es)" in r1.stdout), r1.stdout[-2000:] + r1.stderr[-2000:]       # 220 .. 360 pairs in batches of 100
        files = sorted(os.listdir(o1 / "hla"))
        assert files == sorted(os.listdir(o8 / "hla")) and "R1_bestguess.txt" in files and len(files) == 9
        for fn in files:
            assert (o8 / "hla" / fn).read_bytes() == (o1 / "hla" / fn).read_bytes(), (nm, fn)
        assert (o8 / "reads_per_level.txt").read_bytes() == (o1 / "reads_per_level.txt").read_bytes(), nm
        seen.add((o1 / "hla" / "R1_PP_A_pairs.txt").read_bytes() if (o1 / "hla" / "R1_PP_A_pairs.txt").exists() else (o1 / "reads_per_level.txt").read_bytes())
    assert len(seen) == NS          # eight different samples, not one sample eight times
    # ---- round 6: the same eight samples on ONE listed device -- they take turns on it in sample order, sample k + 1 is decoded on the host while sample k is aligned and
    # typed (HLA-LA.cpp: SampleSchedule) --, with a tail pool of two (hlala_set_tail_pool: three batches in flight per sample): the same bytes again
    outsT = [tmp_path / f"turn_{nm}" for nm in names]
    rt = subprocess.run(args(names, outsT) + ["--devices", "0", "--tailPool", "2"], capture_output=True, text=True, timeout=900, cwd=str(tmp_path))
    assert rt.returncode == 0, rt.stdout[-3000:] + rt.stderr[-3000:]
    assert f"Samples: {NS} on 1 device(s)" in rt.stdout and f"Processed {NS} samples on 1 device(s)" in rt.stdout
    for o8, oT in zip(outs8, outsT):
        for fn in sorted(os.listdir(o8 / "hla")):
            assert (oT / "hla" / fn).read_bytes() == (o8 / "hla" / fn).read_bytes(), (str(oT), fn)
        assert (oT / "reads_per_level.txt").read_bytes() == (o8 / "reads_per_level.txt").read_bytes()
