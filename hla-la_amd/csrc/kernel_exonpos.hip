// kernel_exonpos.hip -- exon positions of the read pairs of a batch for one locus (hla/HLATyper.cpp:1385-1428) on gfx950.
//
// One thread per read pair.  A mate's exon positions come out of one pass over the columns of its selected chain
// (oneReadAlignment_2_exonPositions_paired, :3192-3565): a position per column with a defined level, insertion columns extend the
// genotype of the position before them.  Levels increase strictly along an extended chain, so the two mates' lists are sorted and
// removeDoublePositionsFromRead (:4020-4083: one position per graph level in ascending level order, the alternative with the best
// worst-quality, first wins) is a two-pointer merge of the two streams: nothing is materialised.  Pass 0 counts positions and
// genotype characters per pair, exclusive sums give every accepted pair its slots, pass 1 writes.
#include <hipcub/hipcub.hpp>

#include "batch.h"
#include "../../include/hlala_gpu.h"

namespace hlala {

struct ExonLocus { int level_min, level_max; const int* level_to_exon; double insert_mean, insert_sd, min_mapq, min_weighted_ok; const uint8_t* pair_mask; int min_alignment_columns; };

// one mate as the typer sees it
struct MateAln { int n; const int* lv; const uint8_t* g; const uint8_t* s; const uint8_t* mq; const uint8_t* bases; const uint8_t* quals; int readLen; int first, last; };

// cursor over the exon positions of one mate, in column order
struct PosCursor {
    int j;             // next column to look at
    int seqIdx;        // read bases consumed before column j
    int runF;          // running novel gap of the forward sweep before column j (:3246-3262)
    // the current position (valid after pos_next returned true)
    int col, level, exonPos, nIns, seq0, novelGap, nChars, strip; unsigned char worstQ; bool isGapSeq;
};

__device__ inline void novel_step(unsigned char gc, unsigned char sc, int& run)
{
    if((gc != '_') && (sc != '_')) run = 0;
    else if(!((gc == '_') && (sc == '_'))) run++;
}

// advance to the next position that lies on an exon level; false at the end of the alignment
__device__ inline bool pos_next(const MateAln& a, const ExonLocus& L, PosCursor& c)
{
    while(c.j < a.n) {
        const int j = c.j;
        const unsigned char sc = a.s[j], gc = a.g[j];
        const int lv = a.lv[j];
        novel_step(gc, sc, c.runF);
        const int runFHere = c.runF;
        const int seqHere = c.seqIdx;
        if(sc != '_') c.seqIdx++;
        c.j++;
        if(lv == -1) continue;                                   // an insertion before any position, or one already folded into its position
        // fold the insertion columns that follow (:3299-3340); they advance the sweeps as well
        int nIns = 0;
        while(c.j < a.n && a.lv[c.j] == -1) { novel_step(a.g[c.j], a.s[c.j], c.runF); if(a.s[c.j] != '_') c.seqIdx++; c.j++; nIns++; }
        if(!(lv >= L.level_min && lv <= L.level_max)) continue;
        const int ep = L.level_to_exon[lv - L.level_min];
        if(ep < 0) continue;
        // backward sweep value at column j (:3264-3283): single-gap columns from j up to the next column with two characters
        int runB = 0;
        for(int k = j; k < a.n; k++) { const unsigned char g2 = a.g[k], s2 = a.s[k]; if((g2 != '_') && (s2 != '_')) break; if(!((g2 == '_') && (s2 == '_'))) runB++; }
        c.col = j; c.level = lv; c.exonPos = ep; c.nIns = nIns; c.isGapSeq = (sc == '_'); c.seq0 = seqHere;
        c.novelGap = runFHere > runB ? runFHere : runB;
        c.strip = (c.isGapSeq && nIns > 0) ? 1 : 0;              // "_" + inserted bases: the leading '_' is dropped (:3325-3333)
        c.nChars = 1 + nIns - c.strip;
        // worst quality of the position (0 for a pure "_"), for removeDoublePositionsFromRead
        unsigned char wq = 0; bool have = false;
        const int q0 = c.isGapSeq ? seqHere : seqHere, nQ = (c.isGapSeq ? 0 : 1) + nIns;
        for(int k = 0; k < nQ; k++) { unsigned char q = a.quals[q0 + k]; if(!have || q < wq) { wq = q; have = true; } }
        c.worstQ = (c.isGapSeq && nIns == 0) ? 0 : wq;
        return true;
    }
    return false;
}

// HLATyper::alignmentWeightedOKFraction (:3933-4018) and alignmentFractionOK (:3082-3101); terms are added in column order
__device__ inline void mate_fractions(const MateAln& a, const DevTables& T, double& weightedOK, double& fractionOK, int& colsNonGap)
{
    int idx = -1, ok = 0, checked = 0, nonGap = 0; double weightedMismatches = 0;
    for(int cI = 0; cI < a.n; cI++) {
        const unsigned char sc = a.s[cI], gc = a.g[cI];
        if(sc != '_') {
            nonGap++; idx++;
            if(gc == '_') weightedMismatches++;
            else if(sc != gc) weightedMismatches += T.pcorrect[a.quals[idx]];
        } else if(gc != '_') weightedMismatches++;
        if(!((gc == '_') && (sc == '_'))) { checked++; if(gc == sc) ok++; }
    }
    weightedOK = (1.0 - (weightedMismatches / (double)a.readLen));
    fractionOK = double(ok) / double(checked);
    colsNonGap = nonGap;
}

__device__ inline void emit_position(const MateAln& a, const PosCursor& c, int mate, int q, int ch, const hlala_exon_positions_out& o)
{
    o.pos_exon[q] = c.exonPos; o.pos_level[q] = c.level; o.pos_mate[q] = (uint8_t)mate; o.pos_mapq[q] = a.mq[c.col]; o.pos_novel_gap[q] = c.novelGap;
    o.geno_off[q] = ch;
    if(!c.isGapSeq) { o.geno_chars[ch] = a.s[c.col]; o.qual_chars[ch] = a.quals[c.seq0]; ch++; }
    else if(c.nIns == 0) { o.geno_chars[ch] = '_'; o.qual_chars[ch] = 0; ch++; }
    const int qi0 = c.seq0 + (c.isGapSeq ? 0 : 1);
    for(int k = 0; k < c.nIns; k++) { o.geno_chars[ch] = a.s[c.col + 1 + k]; o.qual_chars[ch] = a.quals[qi0 + k]; ch++; }
}

// PASS 0: cnt[3p] = 1 if the pair yields a read, cnt[3p+1] = positions, cnt[3p+2] = genotype characters; okBroken[0/1] += pair test outcome.
// PASS 1: write the read at slot off[3p], positions from off[3p+1], characters from off[3p+2].
template <int PASS>
__global__ void k_exon_positions(const DevBatch* __restrict__ Bp, const DevTables* __restrict__ Tp, ExonLocus L, int* __restrict__ cnt, const int* __restrict__ off,
                                 int* __restrict__ okBroken, hlala_exon_positions_out o)
{
    const DevBatch& B = *Bp; const DevTables& T = *Tp;
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if(p >= B.n_pairs) return;
    if(PASS == 0) { cnt[3 * p] = 0; cnt[3 * p + 1] = 0; cnt[3 * p + 2] = 0; }
    if(L.pair_mask && !L.pair_mask[p]) return;
    if(B.pair_status[p] != 0) return;
    if(PASS == 1 && cnt[3 * p] == 0) return;
    const int stride = B.stride;
    MateAln a[2];
    if(B.unpaired) {
        // one read per unit: oneReadAlignment_2_exonPositions_unpaired (:3568-3930) and the test of :1476; no removeDoublePositionsFromRead
        const int ch = B.best_chain[p];
        if(ch < 0 || ch >= B.n_chains) return;
        const size_t so = row_base(B, ch);
        MateAln& u = a[0];
        u.n = B.ext_ncols[ch]; u.lv = B.ext_level + so; u.g = B.ext_g + so; u.s = B.ext_s + so; u.mq = B.sel_mapq + (size_t)p * stride;
        u.bases = B.read_bases + B.read_off[p]; u.quals = B.read_quals + B.read_off[p]; u.readLen = B.read_off[p + 1] - B.read_off[p];
        u.first = B.ext_firstlast[4 * ch + 0]; u.last = B.ext_firstlast[4 * ch + 2];
        double w0, f0; int cng0;
        mate_fractions(u, T, w0, f0, cng0);
        const bool readOK = (B.mate_mapq[p] >= L.min_mapq) && (u.n >= L.min_alignment_columns);
        if(PASS == 0) atomicAdd(&okBroken[readOK ? 0 : 1], 1);
        if(!readOK) return;
        const int x1 = u.first, x2 = u.last, y1 = L.level_min, y2 = L.level_max;
        const bool use0 = (x1 != -1) && ((x1 >= y1 && x1 <= y2) || (x2 >= y1 && x2 <= y2) || (y1 >= x1 && y1 <= x2) || (y2 >= x1 && y2 <= x2));
        PosCursor c0; c0.j = 0; c0.seqIdx = 0; c0.runF = 0;
        bool h0 = use0 && pos_next(u, L, c0);
        int nPos = 0, nCh = 0;
        const int slot = PASS == 1 ? off[3 * p] : 0; int q = PASS == 1 ? off[3 * p + 1] : 0, chOff = PASS == 1 ? off[3 * p + 2] : 0;
        if(PASS == 1) o.pos_off[slot] = q;
        while(h0) {
            if(PASS == 1) emit_position(u, c0, 1, q, chOff, o);
            q++; chOff += c0.nChars; nPos++; nCh += c0.nChars;
            h0 = pos_next(u, L, c0);
        }
        if(PASS == 0) { if(nPos > 0) { cnt[3 * p] = 1; cnt[3 * p + 1] = nPos; cnt[3 * p + 2] = nCh; } }
        else {
            o.read_pair[slot] = p; o.read_weighted_ok[2 * slot] = w0; o.read_weighted_ok[2 * slot + 1] = -1;
            o.read_fraction_ok[2 * slot] = f0; o.read_fraction_ok[2 * slot + 1] = -1; o.read_distance[slot] = -1;
            o.read_cols_nongap[2 * slot] = use0 ? cng0 : 0; o.read_cols_nongap[2 * slot + 1] = 0;
            if(o.read_reverse) { o.read_reverse[2 * slot] = B.chain_reverse[ch]; o.read_reverse[2 * slot + 1] = 0; }
            if(o.read_mapq) { o.read_mapq[2 * slot] = B.mate_mapq[p]; o.read_mapq[2 * slot + 1] = -1; }
        }
        return;
    }
    for(int m = 0; m < 2; m++) {
        const int r = 2 * p + m; const int ch = B.best_chain[r];
        if(ch < 0 || ch >= B.n_chains) return;
        const size_t so = row_base(B, ch);
        a[m].n = B.ext_ncols[ch]; a[m].lv = B.ext_level + so; a[m].g = B.ext_g + so; a[m].s = B.ext_s + so; a[m].mq = B.sel_mapq + (size_t)r * stride;
        a[m].bases = B.read_bases + B.read_off[r]; a[m].quals = B.read_quals + B.read_off[r]; a[m].readLen = B.read_off[r + 1] - B.read_off[r];
        a[m].first = B.ext_firstlast[4 * ch + 0]; a[m].last = B.ext_firstlast[4 * ch + 2];
    }
    double w[2], f[2]; int cng[2];
    mate_fractions(a[0], T, w[0], f[0], cng[0]); mate_fractions(a[1], T, w[1], f[1], cng[1]);
    // alignedReadPair_pairsDistanceInGraphLevels, alignerBase.cpp:246-283
    const int dist = (a[0].first < a[1].first) ? (a[1].first - a[0].last - 1) : (a[0].first - a[1].last - 1);
    const bool pairOK = B.strands_valid[p] && (fabs((double)dist - L.insert_mean) <= (5 * L.insert_sd)) && (B.mate_mapq[2 * p] >= L.min_mapq) &&
                        ((w[0] >= L.min_weighted_ok) && (w[1] >= L.min_weighted_ok));                                      // :1404-1410
    if(PASS == 0) atomicAdd(&okBroken[pairOK ? 0 : 1], 1);
    if(!pairOK) return;
    // a mate contributes only if its level range overlaps the exon range (Utilities::intervalsOverlap, Utilities.cpp:168-176, at :3226)
    bool use[2];
    for(int m = 0; m < 2; m++) {
        const int x1 = a[m].first, x2 = a[m].last, y1 = L.level_min, y2 = L.level_max;
        use[m] = (x1 != -1) && ((x1 >= y1 && x1 <= y2) || (x2 >= y1 && x2 <= y2) || (y1 >= x1 && y1 <= x2) || (y2 >= x1 && y2 <= x2));
    }
    PosCursor c0, c1; c0.j = 0; c0.seqIdx = 0; c0.runF = 0; c1 = c0;
    bool h0 = use[0] && pos_next(a[0], L, c0), h1 = use[1] && pos_next(a[1], L, c1);
    int nPos = 0, nCh = 0;
    const int slot = PASS == 1 ? off[3 * p] : 0; int q = PASS == 1 ? off[3 * p + 1] : 0, chOff = PASS == 1 ? off[3 * p + 2] : 0;
    if(PASS == 1) o.pos_off[slot] = q;
    while(h0 || h1) {
        // removeDoublePositionsFromRead: ascending level; on equal levels the better worst-quality, mate 1 on ties (:4049-4068)
        int take;
        if(h0 && h1) { if(c0.level < c1.level) take = 0; else if(c1.level < c0.level) take = 1; else take = (c1.worstQ > c0.worstQ) ? 3 : 2; }
        else take = h0 ? 0 : 1;
        const bool from1 = (take == 1 || take == 3);
        const PosCursor& c = from1 ? c1 : c0;
        if(PASS == 1) emit_position(from1 ? a[1] : a[0], c, from1 ? 2 : 1, q, chOff, o);
        q++; chOff += c.nChars; nPos++; nCh += c.nChars;
        if(take == 0 || take >= 2) h0 = pos_next(a[0], L, c0);
        if(take == 1 || take >= 2) h1 = pos_next(a[1], L, c1);
    }
    if(PASS == 0) { if(nPos > 0) { cnt[3 * p] = 1; cnt[3 * p + 1] = nPos; cnt[3 * p + 2] = nCh; } }
    else {
        o.read_pair[slot] = p; o.read_weighted_ok[2 * slot] = w[0]; o.read_weighted_ok[2 * slot + 1] = w[1];
        o.read_fraction_ok[2 * slot] = f[0]; o.read_fraction_ok[2 * slot + 1] = f[1]; o.read_distance[slot] = dist;
        o.read_cols_nongap[2 * slot] = use[0] ? cng[0] : 0; o.read_cols_nongap[2 * slot + 1] = use[1] ? cng[1] : 0;
        if(o.read_reverse) { o.read_reverse[2 * slot] = B.chain_reverse[B.best_chain[2 * p]]; o.read_reverse[2 * slot + 1] = B.chain_reverse[B.best_chain[2 * p + 1]]; }
        if(o.read_mapq) { o.read_mapq[2 * slot] = B.mate_mapq[2 * p]; o.read_mapq[2 * slot + 1] = B.mate_mapq[2 * p + 1]; }
    }
}

// per-unit alignment statistics (hla/HLATyper.cpp:1043-1097): one thread per pair / read
__global__ void k_unit_stats(const DevBatch* __restrict__ Bp, const DevTables* __restrict__ Tp, hlala_unit_stats_out o)
{
    const DevBatch& B = *Bp; const DevTables& T = *Tp;
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if(p >= B.n_pairs) return;
    const int stride = B.stride; const int nm = B.unpaired ? 1 : 2;
    o.valid[p] = 0; o.strands_valid[p] = 0; o.distance[p] = 0;
    for(int m = 0; m < 2; m++) { o.fraction_ok[2 * p + m] = 0; o.weighted_ok[2 * p + m] = 0; o.n_columns[2 * p + m] = 0; o.mate_mapq[2 * p + m] = 0; }
    if(B.pair_status[p] != 0) return;
    MateAln a[2];
    for(int m = 0; m < nm; m++) {
        const int r = B.unpaired ? p : 2 * p + m; const int ch = B.best_chain[r];
        if(ch < 0 || ch >= B.n_chains) return;
        const size_t so = row_base(B, ch);
        a[m].n = B.ext_ncols[ch]; a[m].lv = B.ext_level + so; a[m].g = B.ext_g + so; a[m].s = B.ext_s + so; a[m].mq = B.sel_mapq + (size_t)r * stride;
        a[m].bases = B.read_bases + B.read_off[r]; a[m].quals = B.read_quals + B.read_off[r]; a[m].readLen = B.read_off[r + 1] - B.read_off[r];
        a[m].first = B.ext_firstlast[4 * ch + 0]; a[m].last = B.ext_firstlast[4 * ch + 2];
    }
    for(int m = 0; m < nm; m++) {
        double w, f; int cng; mate_fractions(a[m], T, w, f, cng);
        o.fraction_ok[2 * p + m] = f; o.weighted_ok[2 * p + m] = w; o.n_columns[2 * p + m] = a[m].n; o.mate_mapq[2 * p + m] = B.mate_mapq[B.unpaired ? p : 2 * p + m];
    }
    if(B.unpaired) { o.fraction_ok[2 * p + 1] = -1; o.weighted_ok[2 * p + 1] = -1; o.mate_mapq[2 * p + 1] = -1; o.distance[p] = -1; }
    else {
        o.strands_valid[p] = B.strands_valid[p];
        o.distance[p] = (a[0].first < a[1].first) ? (a[1].first - a[0].last - 1) : (a[0].first - a[1].last - 1);      // alignerBase.cpp:246-283
    }
    o.valid[p] = 1;
}

}  // namespace hlala
