"""Synthetic PRG + read-pair + BWA-like seed generator (numpy only).

Test/bench infrastructure -- NOT part of the product.  Recipe follows the reference's
`simpleGraphSimulator` (Graph/graphSimulator/simpleGraphSimulator.cpp:20-31, 142-271:
random scaffold, haplotypes mutated at a given density of which 30 % become '_', one
haplotype with Poisson-long gaps) and `simulateBAMAlignments` (:273-367: reads drawn
from the haplotype contigs with their true column alignment), degraded to look like
`bwa mem -a` output (soft-clipped ends so the extension DP has work, secondary
alignments on other haplotypes) as laid out in SURVEY.md section 8(d).

Nothing here reads /root/reference.
"""
from __future__ import annotations

import numpy as np

OPS = "MIDNSHP=X"
OP = {c: i for i, c in enumerate(OPS)}
_COMP = np.zeros(256, dtype=np.uint8)
for a, b in zip(b"ACGTN", b"TGCAN"):
    _COMP[a] = b


def revcomp(a: np.ndarray) -> np.ndarray:
    return _COMP[a[..., ::-1]]


# --------------------------------------------------------------------------- graph

def make_haplotypes(rng, G, n_mut=2, n_largegap=1, mut_density=0.02, gap_frac=0.3,
                    gap_start=0.01, gap_mean=10, extra_identical=0):
    """Aligned haplotypes [H, G] over A,C,G,T,_ (uint8)."""
    nuc = np.frombuffer(b"ACGT", dtype=np.uint8)
    scaffold = nuc[rng.integers(0, 4, G)]
    haps = [scaffold.copy()]
    for _ in range(n_mut):
        h = scaffold.copy()
        ev = rng.random(G) <= mut_density
        isgap = ev & (rng.random(G) < gap_frac)
        issnp = ev & ~isgap
        h[issnp] = nuc[rng.integers(0, 4, int(issnp.sum()))]
        h[isgap] = ord("_")
        haps.append(h)
    for _ in range(n_largegap):
        h = scaffold.copy()
        starts = np.nonzero(rng.random(G) <= gap_start)[0]
        lens = rng.poisson(gap_mean, len(starts))
        last = -1
        for s, l in zip(starts, lens):
            if s <= last or l <= 0:
                continue
            e = min(G - 1, s + l - 1)
            h[s:e + 1] = ord("_")
            last = e
        haps.append(h)
    for i in range(extra_identical):
        haps.append(haps[i % len(haps)].copy())
    H = np.stack(haps)
    # keep the first and last 2 columns gap-free so every contig spans the graph ends
    H[:, :2] = scaffold[:2]
    H[:, -2:] = scaffold[-2:]
    return H


def build_graph(H: np.ndarray, k: int = 1, private=None, force_shared=()):
    """Levelled DAG from aligned haplotypes.  Node classes at level l = haplotypes sharing the
    next k symbols (k = 0: one node per level, parallel edges); node / edge creation order =
    level-major, then first-haplotype order -- this IS the canonical order of the C-ABI.
    private [nh, L] bool (optional): where set, the haplotype has a node of its own at that level."""
    nh, G = H.shape
    L = G + 1
    # class key per (hap, level): polynomial hash of symbols [l, l+k)
    key = np.zeros((nh, L), dtype=np.uint64)
    if k > 0:
        pad = np.concatenate([H, np.zeros((nh, k), dtype=np.uint8)], axis=1).astype(np.uint64)
        for j in range(k):
            key[:, :G] = key[:, :G] * np.uint64(257) + pad[:, j:j + G] + np.uint64(1)
    key[:, G] = 0  # single sink class
    if private is not None:
        own = (np.uint64(1) << np.uint64(60)) + np.arange(nh, dtype=np.uint64)[:, None] * np.ones((1, L), np.uint64)
        key = np.where(private, own, key)
    for l in force_shared:          # one node at this level whatever the next symbols are
        key[:, l] = np.uint64(7)

    def first_occurrence_rank(keys):
        """keys [nh, N] -> (rank [nh, N] of each hap's class in first-hap order, n_classes [N])"""
        n, N = keys.shape
        rep = np.tile(np.arange(n)[:, None], (1, N))
        for h in range(n):
            for h2 in range(h):
                m = (keys[h] == keys[h2]) & (rep[h] == h)
                rep[h][m] = np.minimum(rep[h][m], rep[h2][m])
        is_first = rep == np.arange(n)[:, None]
        cum = np.cumsum(is_first, axis=0) - 1      # rank of a first occurrence
        rank = np.take_along_axis(cum, rep, axis=0)
        return rank.astype(np.int64), is_first.sum(axis=0).astype(np.int64), is_first

    nrank, ncls, _ = first_occurrence_rank(key)
    level_off = np.concatenate([[0], np.cumsum(ncls)]).astype(np.int64)
    n_nodes = int(level_off[-1])
    node_level = np.repeat(np.arange(L), ncls).astype(np.int32)
    node_id = level_off[:-1][None, :] + nrank              # [nh, L]
    # edges: unique (from, label, to) per level in first-hap order
    ekey = (node_id[:, :G].astype(np.uint64) * np.uint64(256) + H.astype(np.uint64)) * np.uint64(1 << 20) \
        + nrank[:, 1:].astype(np.uint64)
    erank, ecls, efirst = first_occurrence_rank(ekey)
    e_off = np.concatenate([[0], np.cumsum(ecls)]).astype(np.int64)
    n_edges = int(e_off[-1])
    edge_id = e_off[:-1][None, :] + erank                   # [nh, G]
    edge_from = np.zeros(n_edges, dtype=np.int32)
    edge_to = np.zeros(n_edges, dtype=np.int32)
    edge_label = np.zeros(n_edges, dtype=np.uint8)
    for h in range(nh):
        m = efirst[h]
        ids = edge_id[h][m]
        edge_from[ids] = node_id[h, :G][m]
        edge_to[ids] = node_id[h, 1:][m]
        edge_label[ids] = H[h][m]
    return dict(n_levels=L, n_nodes=n_nodes, n_edges=n_edges, node_level=node_level,
                edge_from=edge_from, edge_to=edge_to, edge_label=edge_label,
                hap_edge=edge_id.astype(np.int32), hap_node=node_id.astype(np.int32))


def make_contigs(H: np.ndarray):
    """Gap-free haplotype sequences + position->level tables (translation/<id>.txt)."""
    seqs, levels, off = [], [], [0]
    for h in range(H.shape[0]):
        nz = np.nonzero(H[h] != ord("_"))[0]
        seqs.append(H[h][nz])
        levels.append(nz.astype(np.int32))
        off.append(off[-1] + len(nz))
    return dict(n_contigs=H.shape[0], contig_off=np.asarray(off, dtype=np.int64),
                contig_seq=np.concatenate(seqs), contig_level=np.concatenate(levels),
                contig_seqid=np.arange(1, H.shape[0] + 1, dtype=np.int32))


def make_world(seed=1, G=25000, k=1, **kw):
    rng = np.random.default_rng(seed)
    H = make_haplotypes(rng, G, **kw)
    g = build_graph(H, k)
    c = make_contigs(H)
    return dict(H=H, graph=g, contigs=c, G=G)


def make_fan_world(seed=5, G=2400, nh=320, fan=(600, 612), gaps_out=(1200, 150), gaps_in=(2000, 150)):
    """A small world whose nodes have HUNDREDS of edges and gap-path jumps, the way allele-rich levels of a real PRG do (thousands of alleles
    of HLA-B, allele-specific deletions): nh haplotypes with sparse substitutions;
      * fan: every haplotype runs through nodes of its own between levels fan[0] and fan[1] -- one node with nh out-edges, one with nh in-edges;
      * gaps_out = (level, n): n haplotypes carry a deletion that STARTS at that level, each of another length -- their common node there has n
        out-edges labelled '_' and n forward gap-path jumps;
      * gaps_in = (level, n): n other haplotypes carry deletions of different lengths that all END at that level -- n backward jumps at one node."""
    rng = np.random.default_rng(seed)
    nuc = np.frombuffer(b"ACGT", dtype=np.uint8)
    scaffold = nuc[rng.integers(0, 4, G)]
    H = np.tile(scaffold, (nh, 1))
    snp = rng.random((nh, G)) < 0.004
    H[snp] = nuc[rng.integers(0, 4, int(snp.sum()))]
    L = G + 1
    private = np.zeros((nh, L), bool)
    a0, a1 = fan
    H[:, a0:a1] = nuc[rng.integers(0, 4, (nh, a1 - a0))]
    private[:, a0 + 1:a1] = True
    H[:, a0 - 1] = scaffold[a0 - 1]; H[:, a1] = scaffold[a1]; H[:, a1 + 1] = scaffold[a1 + 1]          # one shared node before and after (k = 1: the next symbol decides)
    H[:, a0] = nuc[rng.integers(0, 4, nh)]
    # node(a0) must be ONE node although the haplotypes leave it with different symbols: private=False there and the key of level a0 made equal
    b0, nb = gaps_out
    for i in range(nb):
        ln = 5 + i
        H[i, b0:b0 + ln] = ord("_"); private[i, b0 + 1:b0 + ln] = True
        H[i, b0 - 1] = scaffold[b0 - 1]
    c1, ncg = gaps_in
    for i in range(ncg):
        h = nb + i; ln = 5 + i
        H[h, c1 - ln:c1] = ord("_"); private[h, c1 - ln + 1:c1] = True
    H[:, c1] = scaffold[c1]
    H[:, :2] = scaffold[:2]; H[:, -2:] = scaffold[-2:]
    # shared single nodes at the fan's two ends: the class key of those levels is forced equal for all haplotypes
    g = build_graph(H, 1, private=private, force_shared=(a0, a1))
    c = make_contigs(H)
    return dict(H=H, graph=g, contigs=c, G=G)


# --------------------------------------------------------------------------- reads + seeds

def _cigar(ops):
    return [(int(l) << 4) | OP[o] for l, o in ops if l > 0]


def make_batch(world, n_pairs, seed=3, read_len=150, ins_mean=200.0, ins_sd=35.0, clip_max=30,
               p_secondary=0.5, max_secondary=4, p_random_secondary=0.1, indel_read_frac=0.05,
               qual_lo=2, qual_hi=40, p_no_clip=0.15, hardclip_frac=0.0, p_flip=0.5, haps=None):
    """Read pairs with BAM-like alignment records in the hlala_batch_in layout (haps: sample the pairs from these haplotypes only)."""
    rng = np.random.default_rng(seed)
    C = world["contigs"]
    nh = C["n_contigs"]
    off = C["contig_off"]
    clen = np.diff(off)
    seq = C["contig_seq"]
    lvl = C["contig_level"]
    nuc = np.frombuffer(b"ACGT", dtype=np.uint8)
    G = world["G"]
    # level -> position of the first base at or after that level, per contig
    pos_at_level = []
    for h in range(nh):
        l = lvl[off[h]:off[h + 1]]
        pos_at_level.append(np.searchsorted(l, np.arange(G + 1), side="left"))

    reads_b, reads_q, read_off = [], [], [0]
    chain_off = [0]
    read_primary = []
    ch = dict(contig=[], pos=[], offset=[], AS=[], rev=[], cig=[])
    truth_level0 = []

    # ins_mean/ins_sd describe the INNER distance between the mates (what the reference's pairing
    # step measures: pos_downstream - pos_upstream - 1, alignerBase.cpp:290-329)
    frag = np.maximum(read_len + 10, np.rint(rng.normal(ins_mean, ins_sd, n_pairs)).astype(np.int64) + 2 * read_len)
    hap = rng.integers(0, nh, n_pairs) if haps is None else np.asarray(haps)[rng.integers(0, len(haps), n_pairs)]
    for p in range(n_pairs):
        h = int(hap[p])
        F = int(min(frag[p], clen[h] - 2))
        s = int(rng.integers(0, clen[h] - F))
        flip = rng.random() < p_flip
        for m in range(2):
            upstream = (m == 0) != flip          # which physical mate this read is
            rs = s if upstream else s + F - read_len
            rev = not upstream                   # downstream mate aligns to the reverse strand
            base = off[h] + rs
            # optional indel: read consumes read_len +/- 1 reference bases
            ops_mid = None
            if rng.random() < indel_read_frac:
                at = int(rng.integers(40, read_len - 40))
                if rng.random() < 0.5:           # insertion in read
                    ln = int(rng.integers(1, 4))
                    ref = seq[base:base + read_len - ln]
                    b = np.concatenate([ref[:at], nuc[rng.integers(0, 4, ln)], ref[at:]])
                    ops_mid = ("I", at, ln)
                else:                            # deletion from read
                    ln = int(rng.integers(1, 4))
                    ref = seq[base:base + read_len + ln]
                    b = np.concatenate([ref[:at], ref[at + ln:]])
                    ops_mid = ("D", at, ln)
                if len(b) != read_len:
                    ops_mid = None
            if ops_mid is None:
                b = seq[base:base + read_len].copy()
            # mostly high qualities with a geometric tail and a few uniformly bad bases
            q = np.clip(qual_hi - rng.geometric(0.25, read_len) + 1, qual_lo, qual_hi)
            bad = rng.random(read_len) < 0.03
            q[bad] = rng.integers(qual_lo, qual_hi + 1, int(bad.sum()))
            q = q.astype(np.uint8)
            err = rng.random(read_len) < 10.0 ** (-q.astype(np.float64) / 10.0)
            b = b.copy()
            b[err] = nuc[rng.integers(0, 4, int(err.sum()))]
            # soft clips (Beta(1,4) * clip_max), as in SURVEY 8(d)
            a = 0 if rng.random() < p_no_clip else int(rng.beta(1, 4) * clip_max)
            c = 0 if rng.random() < p_no_clip else int(rng.beta(1, 4) * clip_max)
            if ops_mid is not None:
                a = min(a, ops_mid[1] - 5)
                c = min(c, read_len - ops_mid[1] - ops_mid[2] - 5)
            # primary record
            recs = []
            if ops_mid is None:
                cig = [(a, "S"), (read_len - a - c, "M"), (c, "S")]
            elif ops_mid[0] == "I":
                _, at, ln = ops_mid
                cig = [(a, "S"), (at - a, "M"), (ln, "I"), (read_len - at - ln - c, "M"), (c, "S")]
            else:
                _, at, ln = ops_mid
                cig = [(a, "S"), (at - a, "M"), (ln, "D"), (read_len - at - c, "M"), (c, "S")]
            use_hard = hardclip_frac > 0 and rng.random() < hardclip_frac
            mism = int(err.sum())
            recs.append(dict(contig=h, pos=rs + a, AS=read_len - a - c - 5 * mism, rev=rev, cig=cig, primary=True))
            # secondary alignments: same read placed on other haplotypes at the level of the seed start
            lv0 = int(lvl[base + a])
            if rng.random() < p_secondary:
                nsec = int(min(max_secondary, rng.geometric(0.5)))
                others = rng.permutation(nh)
                for h2 in others[:nsec]:
                    h2 = int(h2)
                    if h2 == h:
                        continue
                    if rng.random() < p_random_secondary:
                        p2 = int(rng.integers(0, clen[h2] - read_len - 1))
                    else:
                        p2 = int(pos_at_level[h2][lv0])
                    a2, c2 = a, c
                    if use_hard:
                        pass
                    mlen = read_len - a2 - c2
                    if p2 + mlen + 1 >= clen[h2]:
                        continue
                    refseg = seq[off[h2] + p2: off[h2] + p2 + mlen]
                    mm = int((refseg != b[a2:a2 + mlen]).sum())
                    recs.append(dict(contig=h2, pos=p2, AS=mlen - 5 * mm, rev=rev,
                                     cig=[(a2, "S"), (mlen, "M"), (c2, "S")], primary=False))
            recs.sort(key=lambda r: -r["AS"])      # AS-descending (processBAM.cpp:1945), stable
            c0 = chain_off[-1]
            for i, r in enumerate(recs):
                if r["primary"]:
                    read_primary.append(c0 + i)
                ch["contig"].append(r["contig"]); ch["pos"].append(r["pos"]); ch["offset"].append(0)
                ch["AS"].append(r["AS"]); ch["rev"].append(1 if r["rev"] else 0); ch["cig"].append(_cigar(r["cig"]))
            chain_off.append(c0 + len(recs))
            reads_b.append(b); reads_q.append(q + 33)
            read_off.append(read_off[-1] + read_len)
            truth_level0.append(lv0)

    cigar_off = np.concatenate([[0], np.cumsum([len(c) for c in ch["cig"]])]).astype(np.int32)
    cigar = np.asarray([x for c in ch["cig"] for x in c], dtype=np.uint32)
    return dict(
        n_pairs=n_pairs,
        read_off=np.asarray(read_off, dtype=np.int32),
        read_bases=np.concatenate(reads_b).astype(np.uint8),
        read_quals=np.concatenate(reads_q).astype(np.uint8),
        chain_off=np.asarray(chain_off, dtype=np.int32),
        read_primary=np.asarray(read_primary, dtype=np.int32),
        n_chains=len(ch["pos"]),
        chain_contig=np.asarray(ch["contig"], dtype=np.int32),
        chain_pos=np.asarray(ch["pos"], dtype=np.int32),
        chain_offset=np.asarray(ch["offset"], dtype=np.int32),
        chain_as=np.asarray(ch["AS"], dtype=np.int32),
        chain_reverse=np.asarray(ch["rev"], dtype=np.uint8),
        cigar_off=cigar_off, cigar=cigar,
        truth_level0=np.asarray(truth_level0, dtype=np.int32),
        insert_mean=float(ins_mean), insert_sd=float(ins_sd),
    )


def make_batch_fast(world, n_pairs, seed=3, read_len=150, ins_mean=200.0, ins_sd=35.0, clip_max=30,
                    p_secondary=0.5, max_secondary=4, p_random_secondary=0.1, qual_lo=2, qual_hi=40,
                    p_no_clip=0.15, p_flip=0.5):
    """Vectorised variant of make_batch for bench-scale inputs (no read indels: CIGARs are S/M/S).
    Same distributions otherwise; used where a million pairs are needed in seconds."""
    rng = np.random.default_rng(seed)
    C = world["contigs"]
    nh = C["n_contigs"]; off = C["contig_off"]; clen = np.diff(off); seq = C["contig_seq"]; lvl = C["contig_level"]
    nuc = np.frombuffer(b"ACGT", dtype=np.uint8)
    G = world["G"]; L = read_len
    if "_pos_at_level" not in world:
        world["_pos_at_level"] = np.stack([np.searchsorted(lvl[off[h]:off[h + 1]], np.arange(G + 1), side="left").astype(np.int32)
                                           for h in range(nh)])
    pal = world["_pos_at_level"]
    hap = rng.integers(0, nh, n_pairs)
    frag = np.maximum(L + 10, np.rint(rng.normal(ins_mean, ins_sd, n_pairs)).astype(np.int64) + 2 * L)
    frag = np.minimum(frag, clen[hap] - 2)
    s = (rng.random(n_pairs) * (clen[hap] - frag)).astype(np.int64)
    flip = rng.random(n_pairs) < p_flip
    n = 2 * n_pairs
    # per read (2p, 2p+1)
    mate = np.tile(np.array([0, 1]), n_pairs)
    rhap = np.repeat(hap, 2)
    upstream = (mate == 0) != np.repeat(flip, 2)
    rs = np.where(upstream, np.repeat(s, 2), np.repeat(s + frag - L, 2))
    rev = ~upstream
    ar = np.arange(L)
    b = seq[(off[rhap] + rs)[:, None] + ar[None, :]].copy()
    q = np.clip(qual_hi - rng.geometric(0.25, (n, L)) + 1, qual_lo, qual_hi)
    bad = rng.random((n, L)) < 0.03
    q[bad] = rng.integers(qual_lo, qual_hi + 1, int(bad.sum()))
    q = q.astype(np.uint8)
    err = rng.random((n, L)) < 10.0 ** (-q.astype(np.float32) / 10.0)
    b[err] = nuc[rng.integers(0, 4, int(err.sum()))]
    a = np.where(rng.random(n) < p_no_clip, 0, (rng.beta(1, 4, n) * clip_max).astype(np.int64))
    c = np.where(rng.random(n) < p_no_clip, 0, (rng.beta(1, 4, n) * clip_max).astype(np.int64))
    mism = err.sum(1)
    lv0 = lvl[off[rhap] + rs + a]
    # ---- records: slot 0 = primary, slots 1.. = secondaries
    rec_read = [np.arange(n)]; rec_contig = [rhap]; rec_pos = [rs + a]; rec_as = [L - a - c - 5 * mism]; rec_prim = [np.ones(n, bool)]
    has_sec = rng.random(n) < p_secondary
    nsec = np.where(has_sec, np.minimum(max_secondary, rng.geometric(0.5, n)), 0)
    for k in range(max_secondary):
        sel = np.nonzero(nsec > k)[0]
        if len(sel) == 0:
            break
        h2 = (rhap[sel] + 1 + rng.integers(0, max(1, nh - 1), len(sel))) % nh
        rnd = rng.random(len(sel)) < p_random_secondary
        p2 = np.where(rnd, (rng.random(len(sel)) * (clen[h2] - L - 2)).astype(np.int64), pal[h2, lv0[sel]].astype(np.int64))
        start = p2 - a[sel]
        ok = (start >= 0) & (start + L + 1 < clen[h2])
        sel, h2, p2, start = sel[ok], h2[ok], p2[ok], start[ok]
        ref2 = seq[(off[h2] + start)[:, None] + ar[None, :]]
        mask = (ar[None, :] >= a[sel][:, None]) & (ar[None, :] < (L - c[sel])[:, None])
        mm = ((ref2 != b[sel]) & mask).sum(1)
        rec_read.append(sel); rec_contig.append(h2); rec_pos.append(p2); rec_as.append(L - a[sel] - c[sel] - 5 * mm); rec_prim.append(np.zeros(len(sel), bool))
    rread = np.concatenate(rec_read); rcontig = np.concatenate(rec_contig); rpos = np.concatenate(rec_pos)
    ras = np.concatenate(rec_as); rprim = np.concatenate(rec_prim)
    slot = np.concatenate([np.full(len(x), i) for i, x in enumerate(rec_read)])
    order = np.lexsort((slot, -ras, rread))            # by read, AS descending, then generation order (stable)
    rread, rcontig, rpos, ras, rprim = rread[order], rcontig[order], rpos[order], ras[order], rprim[order]
    nchains = len(rread)
    chain_off = np.zeros(n + 1, np.int64); np.add.at(chain_off, rread + 1, 1); chain_off = np.cumsum(chain_off)
    read_primary = np.nonzero(rprim)[0]
    # CIGAR: aS (L-a-c)M cS with zero-length ops dropped
    ca, cc = a[rread], c[rread]
    nops = 1 + (ca > 0) + (cc > 0)
    cigar_off = np.concatenate([[0], np.cumsum(nops)])
    cigar = np.zeros(int(cigar_off[-1]), np.uint32)
    pos0 = cigar_off[:-1]
    hasA = ca > 0
    cigar[pos0[hasA]] = (ca[hasA].astype(np.uint32) << 4) | OP["S"]
    mpos = pos0 + hasA
    cigar[mpos] = ((L - ca - cc).astype(np.uint32) << 4) | OP["M"]
    hasC = cc > 0
    cigar[(mpos + 1)[hasC]] = (cc[hasC].astype(np.uint32) << 4) | OP["S"]
    return dict(
        n_pairs=n_pairs, read_off=(np.arange(n + 1) * L).astype(np.int32), read_bases=b.reshape(-1), read_quals=(q + 33).reshape(-1),
        chain_off=chain_off.astype(np.int32), read_primary=read_primary.astype(np.int32), n_chains=int(nchains),
        chain_contig=rcontig.astype(np.int32), chain_pos=rpos.astype(np.int32), chain_offset=np.zeros(nchains, np.int32),
        chain_as=ras.astype(np.int32), chain_reverse=rev[rread].astype(np.uint8),
        cigar_off=cigar_off.astype(np.int32), cigar=cigar, truth_level0=lv0.astype(np.int32),
        insert_mean=float(ins_mean), insert_sd=float(ins_sd))


def make_locus(seed=5, n_clusters=200, exon_length=546, n_reads=300, snp_density=0.03, gap_frac=0.01, read_cover=140, p_unused=0.05):
    """Synthetic HLATyper input for one locus (hlala_exon_in layout): allele clusters that differ at SNP sites from a
    consensus exon string, reads drawn from two of the clusters with quality-dependent errors, occasional deletions
    ('_' genotypes), insertions (genotype length > 1) and host-filtered positions (pos_use = 0)."""
    rng = np.random.default_rng(seed)
    nuc = np.frombuffer(b"ACGT", dtype=np.uint8)
    cons = nuc[rng.integers(0, 4, exon_length)]
    seqs = np.tile(cons, (n_clusters, 1))
    snp = rng.random((n_clusters, exon_length)) < snp_density
    seqs[snp] = nuc[rng.integers(0, 4, int(snp.sum()))]
    gaps = rng.random((n_clusters, exon_length)) < gap_frac
    seqs[gaps] = ord("_")
    truth = rng.integers(0, n_clusters, 2)
    pos_off = [0]; pe, g0, gl, q, use = [], [], [], [], []
    for r in range(n_reads):
        src = seqs[truth[r % 2]]
        start = int(rng.integers(0, max(1, exon_length - read_cover)))
        n = int(min(read_cover, exon_length - start))
        for p in range(start, start + n):
            base = src[p]
            qual = int(np.clip(40 - rng.geometric(0.25) + 1, 2, 40))
            if base != ord("_") and rng.random() < 10 ** (-qual / 10):
                base = nuc[rng.integers(0, 4)]
            if base != ord("_") and rng.random() < 0.002:
                base = ord("_")                                   # deletion in the read
            glen = 1 + (int(rng.integers(1, 4)) if (base != ord("_") and rng.random() < 0.003) else 0)
            pe.append(p); g0.append(base); gl.append(glen); q.append(qual + 33); use.append(0 if rng.random() < p_unused else 1)
        pos_off.append(len(pe))
    return dict(n_clusters=n_clusters, exon_length=exon_length, cluster_seq=seqs.reshape(-1).astype(np.uint8), n_reads=n_reads,
                pos_off=np.asarray(pos_off, np.int32), pos_exon=np.asarray(pe, np.int32), pos_g0=np.asarray(g0, np.uint8),
                pos_glen=np.asarray(gl, np.int32), pos_qual=np.asarray(q, np.uint8), pos_use=np.asarray(use, np.uint8),
                truth=truth)


def as_unpaired(b):
    """View a paired batch as 2 n_pairs single reads (long-read / unpaired mode input: n_pairs = number of reads)."""
    u = dict(b)
    u["n_pairs"] = 2 * int(b["n_pairs"])
    return u


def make_long_batch(world, n_reads, seed=5, len_lo=2000, len_hi=10000, sub=0.05, ins=0.04, dele=0.04, clip_max=60, p_second=0.0, haps=None):
    """Single long reads (BASELINE config 5 style: ONT-like substitution / insertion / deletion rates) with one primary alignment each
    whose CIGAR is derived from the truth (thousands of operations), in the hlala_batch_in layout of an UNPAIRED batch
    (n_pairs = number of reads).  `p_second`: fraction of reads that also get a clean second alignment on another haplotype."""
    rng = np.random.default_rng(seed)
    C = world["contigs"]; nh = C["n_contigs"]; off = C["contig_off"]; clen = np.diff(off); seq = C["contig_seq"]
    nuc = np.frombuffer(b"ACGT", dtype=np.uint8)
    reads_b, reads_q, read_off, chain_off, read_primary = [], [], [0], [0], []
    ch = dict(contig=[], pos=[], offset=[], AS=[], rev=[], cig=[])
    for r in range(n_reads):
        h = int(rng.integers(0, nh)) if haps is None else int(haps[int(rng.integers(0, len(haps)))])
        L = int(rng.integers(len_lo, len_hi + 1)); L = min(L, int(clen[h]) - 10)
        s = int(rng.integers(0, clen[h] - L))
        ref = seq[off[h] + s: off[h] + s + L]
        # walk the reference segment, emitting read bases and run-length encoded operations
        ev = rng.random(L)
        ops = []; rb = []

        def push(o, n=1):
            if ops and ops[-1][1] == o:
                ops[-1][0] += n
            else:
                ops.append([n, o])
        for i in range(L):
            if ev[i] < dele and 0 < i < L - 1:
                push("D")
                continue
            b = ref[i]
            if ev[i] > 1 - sub:
                b = nuc[rng.integers(0, 4)]
            rb.append(b); push("M")
            if rng.random() < ins and i < L - 1:
                k = int(rng.integers(1, 3)); rb.extend(nuc[rng.integers(0, 4, k)]); push("I", k)
        a = int(rng.beta(1, 4) * clip_max); c = int(rng.beta(1, 4) * clip_max)
        head = nuc[rng.integers(0, 4, a)]; tail = nuc[rng.integers(0, 4, c)]
        b_all = np.concatenate([head, np.asarray(rb, np.uint8), tail]).astype(np.uint8)
        cig = [(a, "S")] + [(n, o) for n, o in ops] + [(c, "S")]
        rev = bool(rng.random() < 0.5)
        recs = [dict(contig=h, pos=s, AS=int(len(rb)), rev=rev, cig=cig, primary=True)]
        if rng.random() < p_second and nh > 1:
            h2 = int((h + 1 + rng.integers(0, nh - 1)) % nh)
            mlen = min(len(b_all) - a - c, int(clen[h2]) - 5)
            p2 = int(rng.integers(0, clen[h2] - mlen))
            recs.append(dict(contig=h2, pos=p2, AS=int(mlen // 2), rev=rev, cig=[(a, "S"), (mlen, "M"), (len(b_all) - a - mlen, "S")], primary=False))
        c0 = chain_off[-1]
        for i, rec in enumerate(recs):
            if rec["primary"]:
                read_primary.append(c0 + i)
            ch["contig"].append(rec["contig"]); ch["pos"].append(rec["pos"]); ch["offset"].append(0); ch["AS"].append(rec["AS"])
            ch["rev"].append(1 if rec["rev"] else 0); ch["cig"].append(_cigar(rec["cig"]))
        chain_off.append(c0 + len(recs))
        q = np.clip(20 - rng.geometric(0.3, len(b_all)) + 1, 2, 20).astype(np.uint8)
        reads_b.append(b_all); reads_q.append(q + 33); read_off.append(read_off[-1] + len(b_all))
    cigar_off = np.concatenate([[0], np.cumsum([len(c) for c in ch["cig"]])]).astype(np.int32)
    cigar = np.asarray([x for c in ch["cig"] for x in c], dtype=np.uint32)
    return dict(n_pairs=n_reads, read_off=np.asarray(read_off, np.int32), read_bases=np.concatenate(reads_b).astype(np.uint8),
                read_quals=np.concatenate(reads_q).astype(np.uint8), chain_off=np.asarray(chain_off, np.int32), read_primary=np.asarray(read_primary, np.int32),
                n_chains=len(ch["pos"]), chain_contig=np.asarray(ch["contig"], np.int32), chain_pos=np.asarray(ch["pos"], np.int32),
                chain_offset=np.asarray(ch["offset"], np.int32), chain_as=np.asarray(ch["AS"], np.int32), chain_reverse=np.asarray(ch["rev"], np.uint8),
                cigar_off=cigar_off, cigar=cigar)


def make_long_batch_fast(world, n_reads, seed=5, len_lo=6000, len_hi=14000, sub=0.05, ins=0.04, dele=0.04, clip_max=60, haps=None, starts=None):
    """make_long_batch with the per-base loop vectorised (numpy per read): tens of thousands of DISTINCT reads of ~10 kb in seconds.  Same model: a stretch of
    `len_lo..len_hi` reference bases of a contig, every inner base deleted with probability `dele`, substituted with `sub`, followed by an insertion of one or
    two random bases with `ins`; soft clips Beta(1,4) x clip_max at both ends; one primary alignment whose CIGAR is the truth.  `haps`: contigs to draw from
    (default: those of at least len_hi + 10 bases); `starts`: optional (contig, start) per read instead of random places (e.g. reads laid across gene windows)."""
    rng = np.random.default_rng(seed)
    C = world["contigs"]; off = np.asarray(C["contig_off"], np.int64); clen = np.diff(off); seq = C["contig_seq"]
    if haps is None:
        haps = np.nonzero(clen >= len_hi + 10)[0]
    haps = np.asarray(haps, np.int64)
    nuc = np.frombuffer(b"ACGT", dtype=np.uint8)
    OPC = {"M": 0, "I": 1, "D": 2, "S": 4}
    reads_b, reads_q, cigs = [], [], []
    read_off = np.zeros(n_reads + 1, np.int64); cigar_off = np.zeros(n_reads + 1, np.int64)
    contig = np.zeros(n_reads, np.int32); pos = np.zeros(n_reads, np.int32); AS = np.zeros(n_reads, np.int32); rev = (rng.random(n_reads) < 0.5).astype(np.uint8)
    for r in range(n_reads):
        if starts is not None:
            h, s0 = int(starts[r][0]), int(starts[r][1])
            L = int(min(rng.integers(len_lo, len_hi + 1), clen[h] - s0 - 1))
        else:
            h = int(haps[rng.integers(0, len(haps))]); L = int(min(rng.integers(len_lo, len_hi + 1), clen[h] - 10)); s0 = int(rng.integers(0, clen[h] - L))
        ref = seq[off[h] + s0: off[h] + s0 + L]
        ev = rng.random(L)
        isdel = ev < dele; isdel[0] = False; isdel[-1] = False
        b = ref.copy(); sm = ev > 1 - sub
        b[sm] = nuc[rng.integers(0, 4, int(sm.sum()))]
        k = np.where((rng.random(L) < ins) & ~isdel, rng.integers(1, 3, L), 0); k[-1] = 0
        cnt = np.where(isdel, 0, 1 + k)                               # read bases produced by every reference position
        st = np.cumsum(cnt) - cnt; nb = int(cnt.sum())
        rb = nuc[rng.integers(0, 4, nb)]                               # inserted bases are random; the aligned ones are written over them
        keep = ~isdel
        rb[st[keep]] = b[keep]
        a = int(rng.beta(1, 4) * clip_max); c = int(rng.beta(1, 4) * clip_max)
        b_all = np.concatenate([nuc[rng.integers(0, 4, a)], rb, nuc[rng.integers(0, 4, c)]]).astype(np.uint8)
        # operations in reference order: (D | M) of every position, then its insertion; run-length encoded
        op = np.empty(2 * L, np.int64); ln = np.empty(2 * L, np.int64)
        op[0::2] = np.where(isdel, OPC["D"], OPC["M"]); ln[0::2] = 1
        op[1::2] = OPC["I"]; ln[1::2] = k
        nz = ln > 0; op = op[nz]; ln = ln[nz]
        cut = np.concatenate([[True], op[1:] != op[:-1]]); idx = np.nonzero(cut)[0]
        rl = np.add.reduceat(ln, idx); ro = op[idx]
        cg = (rl.astype(np.uint32) << 4) | ro.astype(np.uint32)
        cg = np.concatenate([np.asarray([(a << 4) | OPC["S"]] if a else [], np.uint32), cg, np.asarray([(c << 4) | OPC["S"]] if c else [], np.uint32)]).astype(np.uint32)
        q = np.clip(20 - rng.geometric(0.3, len(b_all)) + 1, 2, 20).astype(np.uint8) + 33
        reads_b.append(b_all); reads_q.append(q); cigs.append(cg)
        read_off[r + 1] = read_off[r] + len(b_all); cigar_off[r + 1] = cigar_off[r] + len(cg)
        contig[r] = h; pos[r] = s0; AS[r] = int(keep.sum())
    ar = np.arange(n_reads + 1, dtype=np.int32)
    return dict(n_pairs=n_reads, read_off=read_off.astype(np.int32), read_bases=np.concatenate(reads_b), read_quals=np.concatenate(reads_q), chain_off=ar, read_primary=ar[:-1].copy(),
                n_chains=n_reads, chain_contig=contig, chain_pos=pos, chain_offset=np.zeros(n_reads, np.int32), chain_as=AS, chain_reverse=rev,
                cigar_off=cigar_off.astype(np.int32), cigar=np.concatenate(cigs))


def make_long_batches_parallel(world, n_reads, per_batch=10000, seed=700, len_lo=6000, len_hi=14000, procs=10, starts=None):
    """n_reads distinct long reads in batches of per_batch, generated by `procs` child processes side by side (tools/gen_long.py: fresh interpreters that
    never touch the GPU -- the caller may have initialised it).  `starts`: optional [(contig, start)] per read."""
    import shutil
    import subprocess
    import sys
    import tempfile
    d = tempfile.mkdtemp(prefix="hlala_long_")
    try:
        C = world["contigs"]
        np.save(_os.path.join(d, "contig_off.npy"), np.asarray(C["contig_off"], np.int64)); np.save(_os.path.join(d, "contig_seq.npy"), np.asarray(C["contig_seq"], np.uint8))
        jobs = []
        for i, a in enumerate(range(0, n_reads, per_batch)):
            n = min(per_batch, n_reads - a)
            if starts is not None:
                np.save(_os.path.join(d, "starts_%d.npy" % (seed + i)), np.asarray(starts[a:a + n], np.int64))
            jobs.append((seed + i, n, _os.path.join(d, "b%d.npz" % i)))
        out = []; running = []
        def reap(p_, path):
            if p_.wait() != 0:
                raise RuntimeError("tools/gen_long.py failed")
            z = np.load(path); b = {k: z[k] for k in z.files}
            for k in ("n_pairs", "n_chains"):
                b[k] = int(b[k])
            return b
        for sd, n, path in jobs:
            running.append((subprocess.Popen([sys.executable, _os.path.join(_ROOT, "gen_long.py"), d, str(sd), str(n), str(len_lo), str(len_hi), path]), path))
            if len(running) >= procs:
                out.append(reap(*running.pop(0)))
        while running:
            out.append(reap(*running.pop(0)))
        return out
    finally:
        shutil.rmtree(d, ignore_errors=True)


# --------------------------------------------------------------------------- Graph M (tools/graphm/graphm.cpp)

import ctypes as _C
import os as _os
import subprocess as _subprocess

_GM = None
_ROOT = _os.path.dirname(_os.path.abspath(__file__))
QUALITY_MATRIX = _os.path.join(_ROOT, "data", "I101_NA12878.txt")


class _GmParams(_C.Structure):
    _fields_ = [("seed", _C.c_uint64), ("n_levels", _C.c_longlong), ("n_backbone", _C.c_int), ("backbone_div", _C.c_double),
                ("gap_stretch_frac", _C.c_double), ("n_windows", _C.c_int), ("win_len_min", _C.c_int), ("win_len_max", _C.c_int),
                ("alleles_min", _C.c_int), ("alleles_max", _C.c_int), ("exon_site_density", _C.c_double),
                ("intron_site_density", _C.c_double), ("private_rate", _C.c_double), ("suffix_len", _C.c_int),
                ("contigs_per_window", _C.c_int), ("threads", _C.c_int), ("hyper_site_density", _C.c_double)]


class _GmBatchParams(_C.Structure):
    _fields_ = [("seed", _C.c_uint64), ("n_pairs", _C.c_int), ("read_len", _C.c_int), ("jump_mean", _C.c_double), ("jump_sd", _C.c_double),
                ("clip_max", _C.c_int), ("p_no_clip", _C.c_double), ("frac_gene", _C.c_double), ("p_secondary", _C.c_double),
                ("max_secondary", _C.c_int), ("p_random_secondary", _C.c_double), ("p_wrong_strand", _C.c_double), ("p_flip", _C.c_double),
                ("gene_candidates", _C.c_int)]


def _gm():
    global _GM
    if _GM is None:
        so = _os.path.join(_ROOT, "_build", "libgraphm.so")
        src = _os.path.join(_ROOT, "graphm", "graphm.cpp")
        src2 = _os.path.join(_ROOT, "graphm", "bamwriter.cpp")
        if not _os.path.exists(so) or _os.path.getmtime(so) < max(_os.path.getmtime(src), _os.path.getmtime(src2)):
            # (several ranks of one bench may get here at once: one builds, the others wait)
            import fcntl
            _os.makedirs(_os.path.join(_ROOT, "_build"), exist_ok=True)
            with open(_os.path.join(_ROOT, "_build", ".graphm.lock"), "w") as lk:
                fcntl.flock(lk, fcntl.LOCK_EX)
                try:
                    _subprocess.check_call(["make", "-s", "-C", _os.path.join(_ROOT, "graphm")])
                finally:
                    fcntl.flock(lk, fcntl.LOCK_UN)
        L = _C.CDLL(so)
        L.gm_world_create.restype = _C.c_void_p; L.gm_world_create.argtypes = [_C.POINTER(_GmParams)]
        L.gm_world_destroy.argtypes = [_C.c_void_p]
        L.gm_last_error.restype = _C.c_char_p
        vp = _C.c_void_p
        L.gm_world_sizes.argtypes = [vp, vp]; L.gm_world_graph.argtypes = [vp] * 5; L.gm_world_contigs.argtypes = [vp] * 6
        L.gm_world_windows.argtypes = [vp] * 5; L.gm_world_window_matrix.argtypes = [vp, _C.c_int, vp, vp]; L.gm_world_nodes_per_level.argtypes = [vp, vp]
        L.gm_batch_create.restype = vp; L.gm_batch_create.argtypes = [vp, _C.POINTER(_GmBatchParams), _C.c_char_p]
        L.gm_batch_destroy.argtypes = [vp]; L.gm_batch_sizes.argtypes = [vp, vp]; L.gm_batch_get.argtypes = [vp] * 14
        L.bw_open.restype = vp; L.bw_open.argtypes = [_C.c_char_p, _C.c_int, _C.POINTER(_C.c_char_p), vp, _C.c_int, _C.c_int]
        L.bw_append.argtypes = [vp, _C.c_int64] + [vp] * 11; L.bw_close.argtypes = [vp]; L.bw_close.restype = _C.c_longlong
        L.bw_last_error.restype = _C.c_char_p
        _GM = L
    return _GM


# --------------------------------------------------------------------------- BAM files of synthetic samples (tools/graphm/bamwriter.cpp)

class BamWriter:
    """Writes records given as flat arrays into a BAM file (parallel deflate).  refs = [(name, length)]."""

    def __init__(self, path, refs, threads=0, level=1):
        L = _gm()
        names = (_C.c_char_p * max(1, len(refs)))(*[nm.encode() for nm, _ in refs])
        lens = np.array([ln for _, ln in refs], np.int32)
        self.h = L.bw_open(str(path).encode(), len(refs), names, lens.ctypes.data, threads, level)
        if not self.h:
            raise RuntimeError("bw_open: " + L.bw_last_error().decode())
        self.records = 0

    def append(self, name_chars, name_off, flag, ref, pos, cigar_off, cigar, seq_off, bases, quals, as_tag):
        a = [np.ascontiguousarray(x, dt) for x, dt in ((name_chars, np.uint8), (name_off, np.int64), (flag, np.uint16), (ref, np.int32), (pos, np.int32), (cigar_off, np.int64),
                                                      (cigar, np.uint32), (seq_off, np.int64), (bases, np.uint8), (quals, np.uint8), (as_tag, np.int32))]
        n = len(a[2])
        if _gm().bw_append(self.h, n, *[x.ctypes.data for x in a]) != 0:
            raise RuntimeError("bw_append: " + _gm().bw_last_error().decode())
        self.records += n

    def append_batch(self, b, names, order="coordinate", rng=None):
        """All alignments of a batch dict (hlala_batch_in layout, paired) as BAM records: the primary of a mate carries the read (alignment
        orientation, as bwa writes it), the others SEQ '*'; `names` = fixed-width name per pair ([n_pairs, width] uint8).
        order: "coordinate" (reference id, position: what samtools sort leaves), "random" (needs rng) or "given"."""
        nr = 2 * b["n_pairs"]; nc = b["n_chains"]
        chain_off = np.asarray(b["chain_off"], np.int64); read_off = np.asarray(b["read_off"], np.int64); cigar_off = np.asarray(b["cigar_off"], np.int64)
        chain_read = np.repeat(np.arange(nr, dtype=np.int64), np.diff(chain_off))
        is_prim = np.zeros(nc, bool); is_prim[np.asarray(b["read_primary"], np.int64)] = True
        flag = (1 | np.where(chain_read % 2 == 0, 64, 128) | np.where(np.asarray(b["chain_reverse"]) != 0, 16, 0) | np.where(is_prim, 0, 256)).astype(np.uint16)
        pos = (np.asarray(b["chain_pos"], np.int64) + np.asarray(b["chain_offset"], np.int64)).astype(np.int32)
        ref = np.asarray(b["chain_contig"], np.int32)
        if order == "coordinate":
            perm = np.lexsort((pos, ref))
        elif order == "random":
            perm = rng.permutation(nc)
        else:
            perm = np.arange(nc)
        names = np.ascontiguousarray(names, np.uint8); width = names.shape[1]
        pair_of = (chain_read // 2)[perm]
        name_chars = names[pair_of].reshape(-1); name_off = np.arange(nc + 1, dtype=np.int64) * width
        # CIGARs and reads gathered in record order
        clen = np.diff(cigar_off)[perm]; c_off = np.concatenate([[0], np.cumsum(clen)]).astype(np.int64)
        cig_idx = np.repeat(cigar_off[:-1][perm] - c_off[:-1], clen) + np.arange(int(c_off[-1]), dtype=np.int64)
        rlen = np.where(is_prim, (read_off[1:] - read_off[:-1])[chain_read], 0)[perm]; s_off = np.concatenate([[0], np.cumsum(rlen)]).astype(np.int64)
        seq_idx = np.repeat(read_off[:-1][chain_read][perm] - s_off[:-1], rlen) + np.arange(int(s_off[-1]), dtype=np.int64)
        self.append(name_chars, name_off, flag[perm], ref[perm], pos[perm], c_off, np.asarray(b["cigar"], np.uint32)[cig_idx], s_off,
                    np.asarray(b["read_bases"], np.uint8)[seq_idx], np.asarray(b["read_quals"], np.uint8)[seq_idx], np.asarray(b["chain_as"], np.int32)[perm])

    def close(self):
        if self.h:
            n = _gm().bw_close(self.h); self.h = None
            if n < 0:
                raise RuntimeError("bw_close: " + _gm().bw_last_error().decode())
            return n

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def scrambled_names(chunk, n_pairs, prefix=b"r"):
    """Fixed-width read names [n_pairs, width] whose byte order is a permutation of the pair order: prefix, two hex digits of the chunk, eight hex
    digits of (pair index * odd constant mod 2^32).  Returns (names, rank): rank[p] = position of pair p among the chunk's names in name order."""
    scr = (np.arange(n_pairs, dtype=np.uint64) * np.uint64(0x9E3779B1)) & np.uint64(0xFFFFFFFF)
    hexd = np.frombuffer(b"0123456789abcdef", np.uint8)
    cols = [np.full(n_pairs, c, np.uint8) for c in prefix] + [np.full(n_pairs, hexd[(chunk >> 4) & 15], np.uint8), np.full(n_pairs, hexd[chunk & 15], np.uint8)]
    for sh in range(28, -4, -4):
        cols.append(hexd[((scr >> np.uint64(sh)) & np.uint64(15)).astype(np.int64)])
    names = np.stack(cols, axis=1)
    order = np.argsort(scr, kind="stable")
    rank = np.empty(n_pairs, np.int64); rank[order] = np.arange(n_pairs)
    return names, rank


class _GmWorldHandle:
    def __init__(self, h):
        self.h = h

    def __del__(self):
        try:
            if self.h:
                _gm().gm_world_destroy(self.h)
        except Exception:
            pass


def make_world_m(seed=2, n_levels=5_000_000, n_backbone=8, backbone_div=0.003, gap_stretch_frac=0.02, n_windows=40,
                 win_len=(3000, 6000), alleles=(500, 5000), exon_site_density=0.35, intron_site_density=0.04, private_rate=0.001,
                 suffix_len=10, contigs_per_window=24, threads=0, hyper_site_density=0.6):
    """Graph M of SURVEY.md 8(d): backbone haplotypes + gene windows with hundreds to thousands of allele paths, nodes merged
    by the suffix rule of Graph::buildFromHaplotypes (tools/graphm/graphm.cpp).  Same dict layout as make_world, plus `windows`
    (first / last level, alleles, exon columns per window), `nodes_per_level` and the generator handle for make_batch_m."""
    L = _gm()
    p = _GmParams(seed, n_levels, n_backbone, backbone_div, gap_stretch_frac, n_windows, win_len[0], win_len[1], alleles[0], alleles[1],
                  exon_site_density, intron_site_density, private_rate, suffix_len, contigs_per_window, threads, hyper_site_density)
    h = L.gm_world_create(_C.byref(p))
    if not h:
        raise RuntimeError("gm_world_create: " + L.gm_last_error().decode())
    sz = np.zeros(8, np.int64); L.gm_world_sizes(h, sz.ctypes.data)
    nL, N, E, nc, tb, nw, mx, nseg = [int(x) for x in sz]
    node_level = np.zeros(N, np.int32); ef = np.zeros(E, np.int32); et = np.zeros(E, np.int32); el = np.zeros(E, np.uint8)
    L.gm_world_graph(h, node_level.ctypes.data, ef.ctypes.data, et.ctypes.data, el.ctypes.data)
    off = np.zeros(nc + 1, np.int64); seq = np.zeros(tb, np.uint8); lvl = np.zeros(tb, np.int32); cw = np.zeros(nc, np.int32); cr = np.zeros(nc, np.int32)
    L.gm_world_contigs(h, off.ctypes.data, seq.ctypes.data, lvl.ctypes.data, cw.ctypes.data, cr.ctypes.data)
    wf = np.zeros(nw, np.int32); wl = np.zeros(nw, np.int32); wa = np.zeros(nw, np.int32); we = np.zeros(nw, np.int32)
    if nw:
        L.gm_world_windows(h, wf.ctypes.data, wl.ctypes.data, wa.ctypes.data, we.ctypes.data)
    npl = np.zeros(nL, np.int32); L.gm_world_nodes_per_level(h, npl.ctypes.data)
    graph = dict(n_levels=nL, n_nodes=N, n_edges=E, node_level=node_level, edge_from=ef, edge_to=et, edge_label=el)
    contigs = dict(n_contigs=nc, contig_off=off, contig_seq=seq, contig_level=lvl, contig_seqid=np.arange(1, nc + 1, dtype=np.int32))
    return dict(graph=graph, contigs=contigs, G=nL - 1, windows=dict(first_level=wf, last_level=wl, n_alleles=wa, n_exon_cols=we),
                contig_window=cw, contig_row=cr, nodes_per_level=npl, max_nodes_per_level=mx, _gm=_GmWorldHandle(h), kind="graph_m")


def window_matrix(world, k):
    """Aligned allele matrix [alleles, columns] of gene window k and its exon mask."""
    w = world["windows"]; n = int(w["n_alleles"][k]); ln = int(w["last_level"][k] - w["first_level"][k] + 1)
    M = np.zeros((n, ln), np.uint8); ex = np.zeros(ln, np.uint8)
    _gm().gm_world_window_matrix(world["_gm"].h, k, M.ctypes.data, ex.ctypes.data)
    return M, ex


def make_batch_m(world, n_pairs, seed=3, read_len=150, jump_mean=350.0, jump_sd=35.0, clip_max=30, p_no_clip=0.15, frac_gene=0.3,
                 p_secondary=0.5, max_secondary=4, p_random_secondary=0.1, p_wrong_strand=0.01, p_flip=0.5, gene_candidates=6):
    """Read pairs + bwa-like seeds on a Graph M world (hlala_batch_in layout + truth): qualities / errors from the empirical matrix
    tools/data/I101_NA12878.txt stretched to read_len, Poisson indels, start-to-start jump ~ N(jump_mean, jump_sd) (so the inner
    distance the pairing step measures has mean jump_mean - read_len), at least frac_gene of the pairs drawn from allele rows of the
    gene windows.  Extra keys: truth_level [bases] (level of every read base, -1 = inserted), read_window [pairs] (-1 = backbone only)."""
    L = _gm()
    bp = _GmBatchParams(seed, n_pairs, read_len, jump_mean, jump_sd, clip_max, p_no_clip, frac_gene, p_secondary, max_secondary,
                        p_random_secondary, p_wrong_strand, p_flip, gene_candidates)
    h = L.gm_batch_create(world["_gm"].h, _C.byref(bp), QUALITY_MATRIX.encode())
    if not h:
        raise RuntimeError("gm_batch_create: " + L.gm_last_error().decode())
    try:
        sz = np.zeros(4, np.int64); L.gm_batch_sizes(h, sz.ctypes.data)
        nr, nb, nc, ncig = [int(x) for x in sz]
        read_off = np.zeros(nr + 1, np.int32); bases = np.zeros(nb, np.uint8); quals = np.zeros(nb, np.uint8); tl = np.zeros(nb, np.int32)
        rw = np.zeros(n_pairs, np.int32); chain_off = np.zeros(nr + 1, np.int32); prim = np.zeros(nr, np.int32)
        cc = np.zeros(nc, np.int32); cp = np.zeros(nc, np.int32); cas = np.zeros(nc, np.int32); crev = np.zeros(nc, np.uint8)
        cigar_off = np.zeros(nc + 1, np.int32); cigar = np.zeros(ncig, np.uint32)
        L.gm_batch_get(h, read_off.ctypes.data, bases.ctypes.data, quals.ctypes.data, tl.ctypes.data, rw.ctypes.data, chain_off.ctypes.data,
                       prim.ctypes.data, cc.ctypes.data, cp.ctypes.data, cas.ctypes.data, crev.ctypes.data, cigar_off.ctypes.data, cigar.ctypes.data)
    finally:
        L.gm_batch_destroy(h)
    return dict(n_pairs=n_pairs, read_off=read_off, read_bases=bases, read_quals=quals, chain_off=chain_off, read_primary=prim, n_chains=nc,
                chain_contig=cc, chain_pos=cp, chain_offset=np.zeros(nc, np.int32), chain_as=cas, chain_reverse=crev, cigar_off=cigar_off,
                cigar=cigar, truth_level=tl, read_window=rw, insert_mean=float(jump_mean - read_len), insert_sd=float(jump_sd))
