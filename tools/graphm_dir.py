"""A graph DIRECTORY for a Graph M world (test and bench infrastructure): what `HLA-LA --action HLA` reads from --PRG_graph_dir.

  serializedGRAPH                     the flattened arrays in the library's cache format (what `--action prepareGraph` leaves)
  sequences.txt, mapping_PRGonly/referenceGenome.fa, translation/<SequenceID>.txt      the contigs (PRG-only mapping: every row's interval is its
                                      whole FASTA entry PRG_<SequenceID>, mapper/processBAM.cpp:69-88, 1183-1402, 4389-4457)
  PRG/segments.txt + segment files    level names L<level> (Graph::readGraphLoci, Graph/Graph.cpp:2563-2614); for every typed locus one gene window of
                                      the world: its exon runs as <n>_gene_HLA-<locus>_<n>_exon_<2|3>.txt with one row per allele of the window
                                      (hla/HLATyper.cpp:104-214, 1180-1372, 3130-3200), everything else as padding segments.
The BAM reference names of a sample for this directory are PRG_<SequenceID> (ref_names)."""
from __future__ import annotations

import os

import numpy as np

CLASS1 = ("A", "B", "C", "E", "F", "G", "H", "K", "V")        # two exons (exon_2, exon_3), the others one (fill_loci_2_exons, hla/HLATyper.cpp:2812-2846)


def ref_names(world):
    return ["PRG_%d" % int(s) for s in world["contigs"]["contig_seqid"]]


def _exon_runs(ex):
    """maximal runs of exon columns of a window: [(first, last + 1)]"""
    idx = np.nonzero(ex > 0)[0]
    if len(idx) == 0:
        return []
    cuts = np.nonzero(np.diff(idx) > 1)[0]
    starts = np.concatenate([[idx[0]], idx[cuts + 1]]); ends = np.concatenate([idx[cuts], [idx[-1]]]) + 1
    return list(zip(starts.tolist(), ends.tolist()))


def write_graph_dir_m(root, world, pkg, loci=("A", "B", "C", "DQA1", "DQB1", "DRB1"), max_alleles=None):
    """Returns {locus: window index}.  `pkg` = the ctypes binding (for hlala_graph_cache_save)."""
    import ctypes as C
    from tools import synth
    root = str(root)
    for d in ("PRG", "translation", "mapping_PRGonly"):
        os.makedirs(os.path.join(root, d), exist_ok=True)
    lib = C.CDLL(pkg.LIB_PATH)
    pkg.save_graph_cache(lib, world["graph"], os.path.join(root, "serializedGRAPH"))
    # ---- contigs
    ct = world["contigs"]; off = ct["contig_off"]; n = ct["n_contigs"]
    rows = ["SequenceID\tName\tFASTAID\tChr\tStart_1based\tStop_1based"]
    with open(os.path.join(root, "mapping_PRGonly", "referenceGenome.fa"), "wb") as fa:
        for i in range(n):
            sid = int(ct["contig_seqid"][i])
            rows.append("%d\tctg%d\tctg%d\t\t\t" % (sid, i, i))
            fa.write((">PRG_%d\n" % sid).encode()); fa.write(ct["contig_seq"][off[i]:off[i + 1]].tobytes()); fa.write(b"\n")
            lv = ct["contig_level"][off[i]:off[i + 1]]
            with open(os.path.join(root, "translation", "%d.txt" % sid), "w") as tf:
                tf.write("\n".join(map(str, lv.tolist())))                       # (no trailing newline: no extra level-0 position, processBAM.cpp:4406-4412)
    with open(os.path.join(root, "sequences.txt"), "w") as f:
        f.write("\n".join(rows) + "\n")
    # ---- PRG segments: the first exon runs of one window per locus
    wins = world["windows"]; nW = len(wins["first_level"])
    order = np.argsort(wins["first_level"])
    use = {}
    segs = []          # (first level, last level + 1, file name or None for padding, rows)
    k = 0
    for locus in loci:
        while k < nW:
            wI = int(order[k]); k += 1
            M, ex = synth.window_matrix(world, wI)
            runs = _exon_runs(ex)
            need = 2 if locus in CLASS1 else 1
            runs = [r for r in runs if r[1] - r[0] >= 30]
            if len(runs) >= need:
                break
        else:
            raise RuntimeError("not enough gene windows with exon runs for the loci " + ",".join(loci))
        use[locus] = wI
        first = int(wins["first_level"][wI])
        if max_alleles:
            M = M[:max_alleles]
        for j, (a, b) in enumerate(runs[:need]):
            segs.append((first + a, first + b, "gene_HLA-%s_%%d_exon_%d.txt" % (locus, 2 + j), (locus, M[:, a:b])))
    segs.sort(key=lambda s: s[0])
    L = world["graph"]["n_levels"]
    # the graph has L node levels = L - 1 edge levels (level names belong to the edge levels)
    nLev = L - 1
    names = []
    cur = 0; idx = 0
    prg = os.path.join(root, "PRG")

    def level_line(a, b):
        return "IndividualID " + " ".join(["L%d" % x for x in range(a, b)])
    for a, b, fn, payload in segs:
        if a > cur:
            idx += 1; nm = "%d_pad_%d.txt" % (idx, idx)
            with open(os.path.join(prg, nm), "w") as f:
                f.write(level_line(cur, a) + "\n")
            names.append(nm)
        idx += 1; nm = ("%d_" % idx) + (fn % idx)
        locus, M = payload
        with open(os.path.join(prg, nm), "wb") as f:
            f.write((level_line(a, b) + "\n").encode())
            # one row per allele: "<locus>*<5-digit row>:01 c c c ..." -- symbols separated by blanks
            sp = np.full((M.shape[0], 2 * M.shape[1]), ord(" "), np.uint8); sp[:, 1::2] = M
            for r in range(M.shape[0]):
                f.write(("%s*%05d:01" % (locus, r + 1)).encode()); f.write(sp[r].tobytes()); f.write(b"\n")
        names.append(nm)
        cur = b
    if cur < nLev:
        idx += 1; nm = "%d_pad_%d.txt" % (idx, idx)
        with open(os.path.join(prg, nm), "w") as f:
            f.write(level_line(cur, nLev) + "\n")
        names.append(nm)
    with open(os.path.join(prg, "segments.txt"), "w") as f:
        f.write("\n".join(names) + "\n")
    return use
