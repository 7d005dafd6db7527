// kernel_pair.hip -- stage C: pairing + mapping qualities, one wavefront per read pair.
//   pairing loop of processBAM::alignOneReadPair          mapper/processBAM.cpp:3408-3546
//   alignerBase::alignedReadPair_strandsValid             mapper/aligner/alignerBase.cpp:213-244
//   alignedReadPair_pairsDistancesUnderlyingSequences     mapper/aligner/alignerBase.cpp:290-329
//   verboseSeedChain::alignment_{begin,end}_originalSequenceAnchors   mapper/reads/verboseSeedChain.h:230-280
//   processBAM::assignMappingQualities                    mapper/processBAM.cpp:4062-4312
// Insert-size log densities come from a host-built table over integer distances (bit-identical to the host
// libm evaluation); posterior sums are accumulated in the reference's combination order.
#include "batch.h"
#include "../../include/hlala_gpu.h"

namespace hlala {

constexpr int PAIR_CHAINS = 64;     // extended chains per mate
constexpr int PAIR_COMB   = 1024;   // chain combinations per pair
#ifndef HLALA_PAIR_COMB_LDS
#define HLALA_PAIR_COMB_LDS 128      // (tools/gpu_gen_wrap.sh builds with 2 to run the parity tests through the HBM-scratch instance)
#endif
constexpr int PAIR_COMB_LDS = HLALA_PAIR_COMB_LDS;  // ... of which the LDS block holds this many; pairs with more keep theirs in the wave's HBM scratch
constexpr int PAIR_COLS   = 512;    // columns per chain handled by the per-position pass

struct __align__(16) PairLds {
    double LL[PAIR_COMB_LDS];
    int list[2][PAIR_CHAINS];
    int nlist[2];
    short basecol[PAIR_COLS];
    short levcol[PAIR_COLS];
    double red[4];
    int ired[8];
};

// position of sequence `id` at `level` (the level's entries are sorted by sequence id, flat_graph.cpp), -1 if the sequence does not pass through it
__device__ inline int lp_find(const DevGraph& G, int level, int id)
{
    long long lo = G.lp_off[level], hi = G.lp_off[level + 1];
    while(lo < hi) {
        const long long mid = (lo + hi) >> 1; const int v = G.lp_seqid[mid];
        if(v == id) return G.lp_pos[mid];
        if(v < id) lo = mid + 1; else hi = mid;
    }
    return -1;
}

// max_d log pdf(d) over the distances of every underlying sequence both chain ends map to (:3436-3495).  Called by the whole wave with wave-uniform
// arguments: the lanes share the sequences of the upstream level and look each one up in the downstream levels by binary search (a level of a gene
// window carries dozens to thousands of sequences; one lane walking both lists against each other was most of this kernel's time on Graph M).
// The maximum does not depend on the order of the candidates.
__device__ inline double pair_insert_ll(const DevGraph& G, const DevTables& T, const int up0, const int up1, const int dn0, const int dn1)
{
    const int lane = lane_id();
    // upstream chain: last two defined levels (scan order: last, second last); downstream: first two
    const int upL[2] = {up0, up1}, dnL[2] = {dn0, dn1};
    double best = -1.0e300; bool have = false;
    for(int a = 0; a < 2; a++) {
        if(upL[a] < 0) continue;
        const long long lo = G.lp_off[upL[a]], hi = G.lp_off[upL[a] + 1];
        for(long long ia = lo + lane; ia < hi; ia += 64) {
            const int id = G.lp_seqid[ia]; const int endPos = G.lp_pos[ia];
            if(a == 1 && upL[0] >= 0 && lp_find(G, upL[0], id) >= 0) continue;          // first found wins: ids already anchored by the last level
            int beginPos = -1;
            if(dnL[0] >= 0) beginPos = lp_find(G, dnL[0], id);
            if(beginPos < 0 && dnL[1] >= 0) beginPos = lp_find(G, dnL[1], id);
            if(beginPos < 0) continue;
            const long long d = (long long)beginPos - endPos - 1;
            const long long k = d - T.is_dmin;
            const double v = (k >= 0 && k < T.is_n) ? T.is_logpdf[k] : T.is_penalty;      // pdf <= 0 -> penalty (:3447-3464)
            if(!have || v > best) { best = v; have = true; }
        }
    }
    if(!have) best = -1.0e300;
    for(int o = 32; o; o >>= 1) { const double ov = __shfl_xor(best, o); if(ov > best) best = ov; }
    return __ballot(have) ? best : T.is_penalty;
}

// Insert-size estimation (processBAM::estimateInsertSize, processBAM.cpp:1071-1165): for a batch that holds ONE chain per read
// (chain 2p = mate 1, chain 2p + 1 = mate 2 of pair p) the strand test and every distance of
// alignedReadPair_pairsDistancesUnderlyingSequences (alignerBase.cpp:290-329; same anchors as pair_insert_ll).  One thread per pair.
// out_n[p] = -1: a chain is flagged; -2: strands not valid; else the number of distances (duplicates possible, the host builds the set).
constexpr int PAIR_MAXDIST = 32;
__global__ void k_pair_distances(const DevGraph* __restrict__ Gp, const DevBatch* __restrict__ Bp, int* __restrict__ out_n, int* __restrict__ out_d)
{
    const DevGraph& G = *Gp; const DevBatch& B = *Bp;
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if(p >= B.n_pairs) return;
    const int ca = 2 * p, cb = 2 * p + 1;
    if(B.ext_status[ca] != HLALA_CHAIN_OK || B.ext_status[cb] != HLALA_CHAIN_OK) { out_n[p] = -1; return; }
    const int* fa = B.ext_firstlast + 4 * ca; const int* fb = B.ext_firstlast + 4 * cb;
    const bool ra = B.chain_reverse[ca] != 0, rb = B.chain_reverse[cb] != 0;
    bool valid = false;
    if(fa[0] != -1 && fb[0] != -1 && ra != rb) valid = (!ra) ? (fa[0] < fb[0]) : (fa[2] > fb[2]);          // alignerBase.cpp:213-244
    if(!valid) { out_n[p] = -2; return; }
    const int* fl_up = (fa[0] < fb[0]) ? fa : fb; const int* fl_down = (fa[0] < fb[0]) ? fb : fa;           // alignerBase.cpp:294, 312
    int upL[2] = {fl_up[2], fl_up[3]}, dnL[2] = {fl_down[0], fl_down[1]};
    int n = 0;
    for(int a = 0; a < 2; a++) {
        if(upL[a] < 0) continue;
        for(long long ia = G.lp_off[upL[a]]; ia < G.lp_off[upL[a] + 1]; ia++) {
            int id = G.lp_seqid[ia]; int endPos = G.lp_pos[ia];
            if(a == 1 && upL[0] >= 0) {
                bool dup = false;
                for(long long q = G.lp_off[upL[0]]; q < G.lp_off[upL[0] + 1]; q++) if(G.lp_seqid[q] == id) { dup = true; break; }
                if(dup) continue;
            }
            int beginPos = -1;
            for(int b = 0; b < 2 && beginPos < 0; b++) {
                if(dnL[b] < 0) continue;
                for(long long q = G.lp_off[dnL[b]]; q < G.lp_off[dnL[b] + 1]; q++) if(G.lp_seqid[q] == id) { beginPos = G.lp_pos[q]; break; }
            }
            if(beginPos < 0) continue;
            const int d = beginPos - endPos - 1;
            bool seen = false;
            for(int k = 0; k < n && k < PAIR_MAXDIST; k++) if(out_d[(size_t)p * PAIR_MAXDIST + k] == d) seen = true;
            if(!seen) { if(n < PAIR_MAXDIST) out_d[(size_t)p * PAIR_MAXDIST + n] = d; n++; }
        }
    }
    out_n[p] = n;
}

// Everything of a pair after its chain lists are known: combination log likelihoods, first maximum, posteriors, mapping qualities, per-position
// qualities.  LL holds one double per combination: the pair's LDS block for up to PAIR_COMB_LDS combinations (all but a handful of pairs), else the
// wave's scratch in HBM -- the call sites differ in nothing but that pointer, so each instance addresses its memory directly.
template <bool UNPAIRED>
__device__ __forceinline__ void pair_finish(const DevGraph& G, const DevTables& T, const DevBatch& B, PairLds& P, double* __restrict__ LL,
                                            const int p, const int n1, const int n2, const int nComb, const int lane, const int stride)
{
    constexpr int NM = UNPAIRED ? 1 : 2;
        // ---- combination log likelihoods, row-major (i1, i2) (:3408-3506)
        if(UNPAIRED) { for(int i = lane; i < nComb; i += 64) LL[i] = B.ext_ll[P.list[0][i]]; }                         // read1_extendedChains_log_likelihoods, :3743
        else {
            // what a combination needs of its two chains -- first / last two levels, strand, log likelihood -- is read once per CHAIN, lane k holding the
            // k-th chain of either mate (one round trip), not once per combination through wave-uniform loads (three dependent round trips each)
            int4 fA = make_int4(-1, -1, -1, -1), fB = make_int4(-1, -1, -1, -1); int rA = 0, rB = 0; double lA = 0.0, lB = 0.0;
            if(lane < n1) { const int c = P.list[0][lane]; fA = *(const int4*)(B.ext_firstlast + 4 * (size_t)c); rA = B.chain_reverse[c]; lA = B.ext_ll[c]; }
            if(lane < n2) { const int c = P.list[1][lane]; fB = *(const int4*)(B.ext_firstlast + 4 * (size_t)c); rB = B.chain_reverse[c]; lB = B.ext_ll[c]; }
            auto rl64 = [](double v, int l) -> double { const long long b = __double_as_longlong(v);
                return __longlong_as_double((long long)(((u64)(u32)__builtin_amdgcn_readlane((int)(b >> 32), l) << 32) | (u64)(u32)__builtin_amdgcn_readlane((int)b, l))); };
            for(int i = 0; i < nComb; i++) {              // one combination at a time, the wave shares the insert-size term
                const int i1 = i / n2, i2 = i % n2;
                const int fa0 = __builtin_amdgcn_readlane(fA.x, i1), fa1 = __builtin_amdgcn_readlane(fA.y, i1), fa2 = __builtin_amdgcn_readlane(fA.z, i1), fa3 = __builtin_amdgcn_readlane(fA.w, i1);
                const int fb0 = __builtin_amdgcn_readlane(fB.x, i2), fb1 = __builtin_amdgcn_readlane(fB.y, i2), fb2 = __builtin_amdgcn_readlane(fB.z, i2), fb3 = __builtin_amdgcn_readlane(fB.w, i2);
                const bool ra = __builtin_amdgcn_readlane(rA, i1) != 0, rb = __builtin_amdgcn_readlane(rB, i2) != 0;
                bool valid = false;
                if(fa0 != -1 && fb0 != -1 && ra != rb) valid = (!ra) ? (fa0 < fb0) : (fa2 > fb2);          // alignerBase.cpp:213-244
                double llIS = T.is_penalty;
                if(valid) llIS = (fa0 < fb0) ? pair_insert_ll(G, T, fa2, fa3, fb0, fb1) : pair_insert_ll(G, T, fb2, fb3, fa0, fa1);        // alignerBase.cpp:294, 312
                double combined = rl64(lA, i1) + rl64(lB, i2);
                combined += llIS;
                if(lane == 0) LL[i] = combined;
            }
        }
        WSYNC();
        // ---- first maximum (Utilities::findVectorMax, Utilities.cpp:309-323)
        double mx = -1.0e300; int mi = 0x7FFFFFFF;
        for(int i = lane; i < nComb; i += 64) { double v = LL[i]; if(v > mx) { mx = v; mi = i; } }
        for(int o = 32; o; o >>= 1) { double ov = __shfl_xor(mx, o); int oi = __shfl_xor(mi, o); if(ov > mx || (ov == mx && oi < mi)) { mx = ov; mi = oi; } }
        const int bestI = uni(mi), best1 = bestI / n2, best2 = bestI % n2;
        const int selA = uni(P.list[0][best1]), selB = UNPAIRED ? selA : uni(P.list[1][best2]);
        // ---- posterior over combinations (:4064-4085): exp(LL - max), normalised by a left-to-right sum
        double mapQ = 1, q1 = 1, q2 = 1;
        if(nComb > 1) {
            for(int i = lane; i < nComb; i += 64) LL[i] = exp_cr_nonpos(LL[i] - mx);
            WSYNC();
            if(lane == 0) {
                double S = 0; for(int i = 0; i < nComb; i++) S += LL[i];
                P.red[0] = S;
            }
            WSYNC();
            double S = P.red[0];
            for(int i = lane; i < nComb; i += 64) LL[i] = LL[i] / S;
            WSYNC();
            if(lane == 0) {
                double a = 0, b = 0;
                for(int i = 0; i < nComb; i++) { double pp = LL[i]; if(i / n2 == best1) a += pp; if(i % n2 == best2) b += pp; }
                if(a > 1) a = 1; if(b > 1) b = 1;
                P.red[1] = a; P.red[2] = b;
            }
            WSYNC();
            mapQ = LL[bestI]; q1 = P.red[1]; q2 = P.red[2];
        }
        if(lane == 0) {
            B.pair_status[p] = 0; B.n_comb[p] = nComb; B.pair_ll[p] = mx; B.pair_mapq[p] = mapQ;
            if(UNPAIRED) { B.best_chain[p] = selA; B.mate_mapq[p] = mapQ; B.strands_valid[p] = 0; }                       // forReturn.mapQ = mapQ, :3921
            else {
            B.best_chain[2 * p] = selA; B.best_chain[2 * p + 1] = selB; B.mate_mapq[2 * p] = q1; B.mate_mapq[2 * p + 1] = q2;
            const int* fa = B.ext_firstlast + 4 * selA; const int* fb = B.ext_firstlast + 4 * selB;
            bool ra = B.chain_reverse[selA] != 0, rb = B.chain_reverse[selB] != 0; bool valid = false;
            if(fa[0] != -1 && fb[0] != -1 && ra != rb) valid = (!ra) ? (fa[0] < fb[0]) : (fa[2] > fb[2]);
            B.strands_valid[p] = valid ? 1 : 0;
            }
        }
        // ---- per-position mapping quality of the selected chains (:4155-4311)
        for(int m = 0; m < NM; m++) {
            const int sel = m ? selB : selA; const int r = UNPAIRED ? p : 2 * p + m;
            const int nSel = uni(B.ext_ncols[sel]); const size_t sb = (size_t)sel * stride; const size_t ob = (size_t)r * stride;
            if(nComb == 1) {
                unsigned char ph = phred_from_pcorrect(T, 1.0);
                for(int j = lane; j < nSel; j += 64) B.sel_mapq[ob + j] = ph;
                continue;
            }
            const int nl = m ? n2 : n1;
            // columns of the selected chain handled by this lane: j = lane + 64*t
            constexpr int PER = PAIR_COLS / 64;
            u64 mask[PER];
            int myIdx[PER];
            for(int t = 0; t < PER; t++) { mask[t] = 0; myIdx[t] = -1; }
            // read-base ordinal of each selected column (alignment orientation; strand is shared by all chains of the mate)
            {
                int carry = 0;
                for(int t = 0; t < PER; t++) {
                    int j = t * 64 + lane; bool isBase = j < nSel && B.ext_s[sb + j] != '_';
                    u64 bm = __ballot(isBase);
                    if(isBase) myIdx[t] = carry + __popcll(bm & ((1ull << lane) - 1ull));
                    carry += __popcll(bm);
                }
            }
            // level and graph character of the selected chain's columns: read once, not once per other chain
            int selLv[PER]; unsigned char selG[PER];
            for(int t = 0; t < PER; t++) { const int j = t * 64 + lane; selLv[t] = -1; selG[t] = 0; if(j < nSel) { selLv[t] = B.ext_level[sb + j]; selG[t] = B.ext_g[sb + j]; } }
            for(int k = 0; k < nl; k++) {
                const int ck = uni(P.list[m][k]); const int nk = uni(B.ext_ncols[ck]); const size_t kb = (size_t)ck * stride;
                if(ck == sel) {
                    // the selected chain agrees with itself in every column (base columns through their read-base ordinal, gap columns through their level)
                    for(int t = 0; t < PER; t++) if(t * 64 + lane < nSel) mask[t] |= (1ull << k);
                    continue;
                }
                const int firstK = uni(B.ext_firstlast[4 * ck + 0]);
                WSYNC();
                for(int i = lane; i < PAIR_COLS; i += 64) { P.basecol[i] = -1; P.levcol[i] = -1; }
                WSYNC();
                int carry = 0;
                for(int j0 = 0; j0 < nk; j0 += 64) {
                    int j = j0 + lane; bool act = j < nk;
                    bool isBase = act && B.ext_s[kb + j] != '_';
                    u64 bm = __ballot(isBase);
                    if(isBase) { int bi = carry + __popcll(bm & ((1ull << lane) - 1ull)); if(bi < PAIR_COLS) P.basecol[bi] = (short)j; }
                    carry += __popcll(bm);
                    if(act) { int l = B.ext_level[kb + j]; if(l != -1 && firstK >= 0) { int li = l - firstK; if(li >= 0 && li < PAIR_COLS) P.levcol[li] = (short)j; } }
                }
                WSYNC();
                for(int t = 0; t < PER; t++) {
                    int j = t * 64 + lane;
                    if(j >= nSel) continue;
                    const int lj = selLv[t]; const unsigned char gj = selG[t];
                    bool hit = false;
                    if(myIdx[t] >= 0) { int col = P.basecol[myIdx[t]]; if(col >= 0) hit = (B.ext_level[kb + col] == lj) && (B.ext_g[kb + col] == gj); }
                    else if(lj != -1 && firstK >= 0) { int li = lj - firstK; if(li >= 0 && li < PAIR_COLS) { int col = P.levcol[li]; if(col >= 0) hit = (B.ext_s[kb + col] == '_') && (B.ext_g[kb + col] == gj); } }
                    if(hit) mask[t] |= (1ull << k);
                }
            }
            WSYNC();
            for(int t = 0; t < PER; t++) {
                int j = t * 64 + lane;
                if(j >= nSel) continue;
                double Q = 0;                                   // alignmentPositionConfidences accumulated in combination order
                if(m == 0) { for(int i1 = 0; i1 < n1; i1++) if(mask[t] & (1ull << i1)) for(int i2 = 0; i2 < n2; i2++) Q += LL[i1 * n2 + i2]; }
                else       { for(int i1 = 0; i1 < n1; i1++) for(int i2 = 0; i2 < n2; i2++) if(mask[t] & (1ull << i2)) Q += LL[i1 * n2 + i2]; }
                if(Q > 1) Q = 1;
                B.sel_mapq[ob + j] = phred_from_pcorrect(T, Q);
            }
        }
}

// UNPAIRED: one read per unit (processBAM::alignOneLongRead :3618-3838 selects the first maximum of the chains' log likelihoods;
// assignMappingQualities_unpaired :3900-4059 is the paired computation with a single, neutral second mate).
template <bool UNPAIRED>
__global__ __launch_bounds__(64, 5) void k_pair_chains(const DevGraph* __restrict__ Gp, const DevTables* __restrict__ Tp, const DevBatch* __restrict__ Bp,
                                                   const uint8_t* __restrict__ deferPairs, const int deferMode, const int counterIdx,      // deferMode 1: skip deferred pairs, 2: only those
                                                   double* __restrict__ bigLL)                    // [gridDim.x][PAIR_COMB]
{
    const DevGraph& G = *Gp;
    const DevBatch& B = *Bp;
    __shared__ PairLds P;
    const int lane = lane_id();
    const DevTables& T = *Tp;
    const int stride = B.stride;

    // pairs are drawn eight at a time (one same-address atomic per pair serialises the grid at the L2)
    constexpr int CHUNK = 8;
    int sweep = 0;
    for(;;) {
        int p0 = 0;
        if(deferMode == 2) { p0 = ((int)blockIdx.x + sweep * (int)gridDim.x) * CHUNK; sweep++; }      // second pass (a few thousand pairs of a million): the waves sweep the flags, no draws
        else {
            if(lane == 0) p0 = atomicAdd(&B.work_counter[counterIdx], CHUNK);
            p0 = __builtin_amdgcn_readfirstlane(p0);
        }
        if(p0 >= B.n_pairs) break;
        const int pEnd = min(p0 + CHUNK, B.n_pairs);
        // the chunk's deferred flags and chain ranges: lane q holds pair p0 + q (one round trip for the chunk, not two dependent ones per pair)
        constexpr int NM = UNPAIRED ? 1 : 2;
        int hDf = 0, hC[NM + 1];
        #pragma unroll
        for(int m = 0; m <= NM; m++) hC[m] = 0;
        if(lane < CHUNK && p0 + lane < pEnd) {
            const int pq = p0 + lane;
            if(deferMode) hDf = deferPairs[pq];
            #pragma unroll
            for(int m = 0; m <= NM; m++) hC[m] = B.chain_off[NM * pq + m];
        }
        for(int p = p0; p < pEnd; p++) {
        const int hq = p - p0;
        if(deferMode) { const bool df = __builtin_amdgcn_readlane(hDf, hq) != 0; if(df == (deferMode == 1)) continue; }
        // ---- lists of extended chains per mate (read1_extendedChains / read2_extendedChains), error propagation
        int bad = 0;
        int cLo[NM], cHi[NM], stF[NM], ncF[NM];
        #pragma unroll
        for(int m = 0; m < NM; m++) { cLo[m] = __builtin_amdgcn_readlane(hC[m], hq); cHi[m] = __builtin_amdgcn_readlane(hC[m + 1], hq); }
        // status and length of the first 64 chains of both mates: one round trip (a pair with more alignments per mate loops on)
        #pragma unroll
        for(int m = 0; m < NM; m++) { const int c = cLo[m] + lane; stF[m] = 1; ncF[m] = 0; if(c < cHi[m]) { stF[m] = B.ext_status[c]; ncF[m] = B.ext_ncols[c]; } }
        int mxc = 0;                 // longest listed chain: the per-position pass below holds PAIR_COLS columns per chain (long reads come with one alignment each: nComb == 1, any length)
        #pragma unroll
        for(int m = 0; m < NM; m++) {
            const int c0 = cLo[m], c1 = cHi[m];
            int cnt = 0;
            for(int b0 = c0; b0 < c1; b0 += 64) {
                int c = b0 + lane; int st, nc;
                if(b0 == c0) { st = stF[m]; nc = ncF[m]; } else { st = 1; nc = 0; if(c < c1) { st = B.ext_status[c]; nc = B.ext_ncols[c]; } }
                if(__ballot(st < 0)) bad = 1;
                u64 okm = __ballot(st == HLALA_CHAIN_OK);
                if(st == HLALA_CHAIN_OK) { int pos = cnt + __popcll(okm & ((1ull << lane) - 1ull)); if(pos < PAIR_CHAINS) { P.list[m][pos] = c; mxc = max(mxc, nc); } }
                cnt += __popcll(okm);
            }
            if(lane == 0) P.nlist[m] = cnt;
            if(cnt < 1 || cnt > PAIR_CHAINS) bad = 1;
        }
        WSYNC();
        const int n1 = uni(P.nlist[0]), n2 = UNPAIRED ? 1 : uni(P.nlist[1]);
        bad = uni(bad);
        const long long nCombLL = (long long)n1 * n2;
        if(!bad && nCombLL > PAIR_COMB) bad = 1;
        mxc = wave_max_i32(mxc);
        if(!bad && nCombLL > 1 && mxc > PAIR_COLS) bad = 1;
        if(bad) {
            if(lane == 0) { B.pair_status[p] = -1; if(UNPAIRED) B.best_chain[p] = -1; else { B.best_chain[2 * p] = -1; B.best_chain[2 * p + 1] = -1; } B.n_comb[p] = 0; }
        } else {
        const int nComb = (int)nCombLL;
        if(nComb <= PAIR_COMB_LDS) pair_finish<UNPAIRED>(G, T, B, P, P.LL, p, n1, n2, nComb, lane, stride);
        else pair_finish<UNPAIRED>(G, T, B, P, bigLL + (size_t)blockIdx.x * PAIR_COMB, p, n1, n2, nComb, lane, stride);
        }   // !bad
        WSYNC();
        }
    }
}

// Per-pair post-processing (processBAM.cpp:2411-2446): coverage counters over the columns of both selected chains and the
// includeInHLA flag (closed-interval overlap of [first level, last level] of a mate with a gene interval).  One wave per pair.
__global__ __launch_bounds__(64) void k_post_pairs(const DevBatch* __restrict__ Bp, int* __restrict__ cov, int nCov, const int* __restrict__ geneFirst,
                                                   const int* __restrict__ geneLast, int nGenes, uint8_t* __restrict__ include)
{
    const DevBatch& B = *Bp;
    const int lane = lane_id();
    const int stride = B.stride;
    constexpr int CHUNK = 8;
    for(;;) {
        int p0 = 0;
        if(lane == 0) p0 = atomicAdd(&B.work_counter[11], CHUNK);
        p0 = __builtin_amdgcn_readfirstlane(p0);
        if(p0 >= B.n_pairs) break;
        const int pEnd = min(p0 + CHUNK, B.n_pairs);
        for(int p = p0; p < pEnd; p++) {
            bool inc = false;
            if(uni(B.pair_status[p]) == 0) {
                const int nm = B.unpaired ? 1 : 2;
                for(int m = 0; m < nm; m++) {
                    const int ch = uni(B.best_chain[B.unpaired ? p : 2 * p + m]);
                    if(ch < 0 || ch >= B.n_chains) continue;
                    const int n = uni(B.ext_ncols[ch]);
                    const size_t so = (size_t)ch * stride;
                    for(int j = lane; j < n; j += 64) {
                        const int lv = B.ext_level[so + j];
                        if(lv != -1 && B.ext_g[so + j] != '_' && lv >= 0 && lv < nCov) atomicAdd(&cov[lv], 1);          // :2414-2418
                    }
                    const int first = uni(B.ext_firstlast[4 * ch + 0]), last = uni(B.ext_firstlast[4 * ch + 2]);          // alignment_firstLevel / _lastLevel
                    if(first != -1) {
                        bool hit = false;
                        for(int g = lane; g < nGenes; g += 64) hit = hit || (geneLast[g] >= first && geneFirst[g] <= last);      // IntervalTree.h:166
                        if(__ballot(hit)) inc = true;
                    }
                }
            }
            if(lane == 0 && include) include[p] = inc ? 1 : 0;
        }
    }
}

// Columns of the selected chains, gathered into read-major staging rows for one bulk copy per array
// (hlala_batch_get_pairs): read r0 + blockIdx.x -> row blockIdx.x; columns beyond n_cols are zero.
__global__ void k_gather_selected(const DevBatch* __restrict__ Bp, int r0, int nRows, int* __restrict__ oN, int* __restrict__ oLevel, int* __restrict__ oEdge,
                                  uint8_t* __restrict__ oG, uint8_t* __restrict__ oS, uint8_t* __restrict__ oFs)
{
    const DevBatch& B = *Bp;
    const int row = blockIdx.x;
    if(row >= nRows) return;
    const int r = r0 + row;
    const int stride = B.stride;
    const int ch = B.best_chain[r];
    int n = 0;
    if(ch >= 0 && ch < B.n_chains) n = B.ext_ncols[ch];
    if(n < 0) n = 0;
    if(threadIdx.x == 0) oN[row] = n;
    const size_t so = (size_t)(ch >= 0 ? ch : 0) * stride, dofs = (size_t)row * stride;
    for(int j = threadIdx.x; j < stride; j += blockDim.x) {
        const bool in = j < n;
        oLevel[dofs + j] = in ? B.ext_level[so + j] : 0; oEdge[dofs + j] = in ? B.ext_edge[so + j] : 0;
        oG[dofs + j] = in ? B.ext_g[so + j] : 0; oS[dofs + j] = in ? B.ext_s[so + j] : 0; oFs[dofs + j] = in ? B.ext_fromseed[so + j] : 0;
    }
}

// packed export of the selected alignments: column counts per read, then one block per read copies its columns to its offset
__global__ void k_selected_ncols(const DevBatch* __restrict__ Bp, int nReads, long long* __restrict__ n)
{
    const DevBatch& B = *Bp;
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if(r > nReads) return;
    long long v = 0;
    if(r < nReads) { const int ch = B.best_chain[r]; if(ch >= 0 && ch < B.n_chains) { const int c = B.ext_ncols[ch]; if(c > 0) v = c; } }
    n[r] = v;
}
__global__ void k_gather_packed(const DevBatch* __restrict__ Bp, int nReads, const long long* __restrict__ off, int* __restrict__ oLevel, int* __restrict__ oEdge,
                                uint8_t* __restrict__ oG, uint8_t* __restrict__ oS, uint8_t* __restrict__ oFs, uint8_t* __restrict__ oMq)
{
    const DevBatch& B = *Bp;
    for(int r = blockIdx.x; r < nReads; r += gridDim.x) {
        const long long d0 = off[r]; const int n = (int)(off[r + 1] - d0);
        if(n <= 0) continue;
        const size_t so = (size_t)B.best_chain[r] * B.stride, mo = (size_t)r * B.stride;
        for(int j = threadIdx.x; j < n; j += blockDim.x) {
            if(oLevel) oLevel[d0 + j] = B.ext_level[so + j];
            if(oEdge) oEdge[d0 + j] = B.ext_edge[so + j];
            if(oG) oG[d0 + j] = B.ext_g[so + j];
            if(oS) oS[d0 + j] = B.ext_s[so + j];
            if(oFs) oFs[d0 + j] = B.ext_fromseed[so + j];
            if(oMq) oMq[d0 + j] = B.sel_mapq[mo + j];
        }
    }
}

__global__ void k_export_pairs(const DevBatch* __restrict__ Bp, double* out)
{
    const DevBatch& B = *Bp;
    int p = blockIdx.x * blockDim.x + threadIdx.x;
    if(p >= B.n_pairs) return;
    double* o = out + 8 * (size_t)p;
    if(B.unpaired) { o[0] = B.pair_status[p]; o[1] = B.best_chain[p]; o[2] = -1; o[3] = B.n_comb[p]; o[4] = B.pair_ll[p]; o[5] = B.pair_mapq[p]; o[6] = B.mate_mapq[p]; o[7] = 0; return; }
    o[0] = B.pair_status[p]; o[1] = B.best_chain[2 * p]; o[2] = B.best_chain[2 * p + 1]; o[3] = B.n_comb[p];
    o[4] = B.pair_ll[p]; o[5] = B.pair_mapq[p]; o[6] = B.mate_mapq[2 * p]; o[7] = B.mate_mapq[2 * p + 1];
}

}  // namespace hlala
