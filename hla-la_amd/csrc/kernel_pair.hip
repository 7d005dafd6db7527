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
#ifndef PAIR_LEAN_WAVES
#define PAIR_LEAN_WAVES 6           // waves per SIMD k_pair_chains / k_pair_multi<., false> are compiled for (tools/gpu_r6_pair.sh measures the alternatives)
#endif
#ifndef PAIR_MULTI_WAVES
#define PAIR_MULTI_WAVES 4
#endif
constexpr int PAIR_COLS   = 512;    // columns per chain handled by the per-position pass

// Per-position pass: what the columns of the chain under comparison look like from the selected chain's side -- by read-base ordinal and by level
// (level - the chain's first level; a chain's defined levels rise by one per column, so 512 columns span fewer than 512 levels)
constexpr short PAIR_NOCOL = -32768;     // no base column with this ordinal
constexpr short PAIR_NOLEVEL = -32767;   // the column carries no level (-1: a base the read inserts)
// the chain lists of a pair: all that k_pair_chains itself keeps in LDS (0.7 KB per wavefront; round 6 -- the tables of the multi-combination pairs live in k_pair_multi)
struct __align__(16) PairListLds {
    int list[2][PAIR_CHAINS];
    int nlist[2];
    unsigned char src[2][PAIR_CHAINS];   // the lane that loaded the facts of list entry k (PairChain)
};
struct __align__(16) PairLds {
    PairListLds L;
    double LL[PAIR_COMB_LDS];
    double phredThr[256];                // DevTables::phred_thr: the binary search of PCorrectToPhred runs on this copy (eight dependent loads per column otherwise)
    short baseLev[PAIR_COLS];            // [read-base ordinal] relative level of that base's column (PAIR_NOCOL / PAIR_NOLEVEL)
    unsigned char baseG[PAIR_COLS];      // ... its graph character
    unsigned char levS[PAIR_COLS];       // [relative level] read character of the column on that level (0: none)
    unsigned char levG[PAIR_COLS];       // ... its graph character
    double red[4];
    int ired[8];
};

// Utilities::PCorrectToPhred (Utilities.cpp:178-203), device_common.h: phred_from_pcorrect on the block's copy of the threshold table
__device__ __forceinline__ unsigned char pair_phred(const PairLds& P, double pCorrect)
{
    double pWrong = 1 - pCorrect;
    if(pWrong == 0) pWrong = 1e-100;
    int lo = 0, hi = 255;
    while(lo < hi) { const int mid = (lo + hi + 1) >> 1; if(pWrong <= P.phredThr[mid]) lo = mid; else hi = mid - 1; }
    return (unsigned char)lo;
}

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
    // Levels of the backbone carry a handful of sequences.  When none of the four levels has more than 16, a quarter of the wave takes each level: the four
    // (offset, count) pairs in one round trip, every (sequence, position) entry in a second, the look-ups between the quarters through lane reads, the table value
    // in a third -- the general form below spends a dependent load per step of every binary search (twenty round trips per combination).
    {
        const int g = lane >> 4, e = lane & 15;
        const int lvl = g == 0 ? up0 : (g == 1 ? up1 : (g == 2 ? dn0 : dn1));
        long long lo = 0; int n = 0;
        if(lvl >= 0) { lo = G.lp_off[lvl]; n = (int)(G.lp_off[lvl + 1] - lo); }
        if(__ballot(n > 16) == 0) {
            int id = -1, pos = -1;
            if(e < n) { id = G.lp_seqid[lo + e]; pos = G.lp_pos[lo + e]; }
            const int nU0 = __builtin_amdgcn_readlane(n, 0), nD0 = __builtin_amdgcn_readlane(n, 32), nD1 = __builtin_amdgcn_readlane(n, 48);
            bool cand = g < 2 && e < n;
            for(int u = 0; u < nU0; u++) {          // first found wins: ids already anchored by the last level
                const int oid = __builtin_amdgcn_readlane(id, u), op = __builtin_amdgcn_readlane(pos, u);
                if(g == 1 && oid == id && op >= 0) cand = false;
            }
            int beginPos = -1;
            for(int u = 0; u < nD0; u++) {
                const int oid = __builtin_amdgcn_readlane(id, 32 + u), op = __builtin_amdgcn_readlane(pos, 32 + u);
                if(oid == id && beginPos < 0) beginPos = op;
            }
            for(int u = 0; u < nD1; u++) {
                const int oid = __builtin_amdgcn_readlane(id, 48 + u), op = __builtin_amdgcn_readlane(pos, 48 + u);
                if(oid == id && beginPos < 0) beginPos = op;
            }
            const bool have = cand && beginPos >= 0;
            double best = -1.0e300;
            if(have) {
                const long long d = (long long)beginPos - pos - 1;
                const long long k = d - T.is_dmin;
                best = (k >= 0 && k < T.is_n) ? T.is_logpdf[k] : T.is_penalty;      // pdf <= 0 -> penalty (:3447-3464)
            }
            for(int o = 32; o; o >>= 1) { const double ov = __shfl_xor(best, o); if(ov > best) best = ov; }
            return __ballot(have) ? best : T.is_penalty;
        }
    }
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
// What the pairing needs of one extended chain -- first / last two levels, strand, log likelihood, length, column row --, lane k holding the k-th chain of a
// mate's list.  Every field of every candidate chain of the pair is requested in ONE round trip, beside the status that decides whether the chain is listed at
// all, and moved to the list's lane order through lane permutes: each dependent load of this kernel is a random 64-byte access into arrays of tens of megabytes
// (1-2 us with the TLB miss), and there used to be four in a row before the first combination was scored (status -> list -> levels / strand / likelihood;
// later length / row / first level again for the per-position pass, and the selected chains' levels once more for the pair's outputs).
struct PairChain { int4 fl; int rev, nk, row; double ll; };
__device__ __forceinline__ PairChain pair_chain_load(const DevBatch& B, const int c)
{
    PairChain x;
    x.fl = *(const int4*)(B.ext_firstlast + 4 * (size_t)c); x.rev = B.chain_reverse[c]; x.ll = B.ext_ll[c]; x.nk = B.ext_ncols[c]; x.row = B.chain_row ? B.chain_row[c] : c;
    return x;
}
__device__ __forceinline__ PairChain pair_chain_from_lane(const PairChain& x, const int srcLane)
{
    PairChain y;
    y.fl.x = __shfl(x.fl.x, srcLane); y.fl.y = __shfl(x.fl.y, srcLane); y.fl.z = __shfl(x.fl.z, srcLane); y.fl.w = __shfl(x.fl.w, srcLane);
    y.rev = __shfl(x.rev, srcLane); y.nk = __shfl(x.nk, srcLane); y.row = __shfl(x.row, srcLane);
    const long long b = __double_as_longlong(x.ll);
    y.ll = __longlong_as_double((long long)(((u64)(u32)__shfl((int)(b >> 32), srcLane) << 32) | (u64)(u32)__shfl((int)b, srcLane)));
    return y;
}

// Per-position mapping quality of the selected chains (:4155-4311).  PER: columns per lane (3: chains of up to 192 columns -- 2x150 bp reads --, 8: up to PAIR_COLS).
// ---- per-position mapping quality of the selected chains (:4155-4311)
// A column of the selected chain is "shared" by another chain of the mate when that chain aligns the same read base to the same level and graph character
// (base columns, matched through their read-base ordinal) or carries a gap with the same graph character on the same level (gap columns, matched through
// the level).  Per other chain: its columns in ONE round trip (up to 256 per lane round), turned into two LDS tables -- by base ordinal and by relative
// level -- that hold what the comparison needs, so the comparison itself reads LDS only.  (Rounds 1-4: column numbers in LDS, the compared fields read back
// from HBM per column, the chain's length / row / first level through three dependent wave-uniform loads, and PCorrectToPhred's binary search on the
// table in global memory: ~75 dependent round trips for a pair of two chains per mate, 85 k cycles.)
template <bool UNPAIRED, int PER>
__device__ __forceinline__ void pair_positions(const DevBatch& B, PairLds& P, const double* __restrict__ LL, const int p, const int n1, const int n2, const int nComb,
                                               const int best1, const int best2, const PairChain& cA, const PairChain& cB, const int lane, const int stride)
{
    constexpr int NM = UNPAIRED ? 1 : 2;
    for(int m = 0; m < NM; m++) {
        const int r = UNPAIRED ? p : 2 * p + m;
        const int nl = m ? n2 : n1;
        // length, column row and first level of every chain of the mate's list: lane k holds chain k (PairChain)
        const int nkL = m ? cB.nk : cA.nk, rowL = m ? cB.row : cA.row, f0L = m ? cB.fl.x : cA.fl.x;
        const int kSel = m ? best2 : best1;
        const int nSel = __builtin_amdgcn_readlane(nkL, kSel); const size_t sb = (size_t)__builtin_amdgcn_readlane(rowL, kSel) * (size_t)stride; const size_t ob = (size_t)r * stride;
        if(nComb == 1) {
            const unsigned char ph = (unsigned char)P.ired[7];          // PCorrectToPhred(1), set when the block started
            for(int j = lane; j < nSel; j += 64) B.sel_mapq[ob + j] = ph;
            continue;
        }
        // columns of the selected chain handled by this lane: j = lane + 64*t
        u64 mask[PER];
        int selLv[PER], selInfo[PER];          // selInfo: graph character | base ordinal << 8 (0xFFFF: a gap column) | read character << 24 (until the ordinals are known)
        #pragma unroll
        for(int t = 0; t < PER; t++) { const int j = t * 64 + lane; mask[t] = 0; selLv[t] = -1; selInfo[t] = (int)((u32)'_' << 24); if(j < nSel) { selLv[t] = B.ext_level[sb + j]; selInfo[t] = (int)((u32)B.ext_g[sb + j] | ((u32)B.ext_s[sb + j] << 24)); } }
        // read-base ordinal of each selected column (alignment orientation; strand is shared by all chains of the mate)
        {
            int carry = 0;
            #pragma unroll
            for(int t = 0; t < PER; t++) {
                if(t * 64 >= nSel) break;
                const int j = t * 64 + lane; const bool isBase = j < nSel && ((u32)selInfo[t] >> 24) != (u32)'_';
                const u64 bm = __ballot(isBase);
                const int ord = isBase ? carry + __popcll(bm & ((1ull << lane) - 1ull)) : 0xFFFF;
                selInfo[t] = (int)(((u32)selInfo[t] & 255u) | ((u32)ord << 8));
                carry += __popcll(bm);
            }
            #pragma unroll
            for(int t = 0; t < PER; t++) if(t * 64 >= nSel) selInfo[t] = (int)(((u32)selInfo[t] & 255u) | (0xFFFFu << 8));
        }
        for(int k = 0; k < nl; k++) {
            if(k == kSel) {
                // the selected chain agrees with itself in every column (base columns through their read-base ordinal, gap columns through their level)
                #pragma unroll
                for(int t = 0; t < PER; t++) if(t * 64 + lane < nSel) mask[t] |= (1ull << k);
                continue;
            }
            const int nk = __builtin_amdgcn_readlane(nkL, k), firstK = __builtin_amdgcn_readlane(f0L, k);
            const size_t kb = (size_t)__builtin_amdgcn_readlane(rowL, k) * (size_t)stride;
            WSYNC();
            for(int i = lane; i < PAIR_COLS / 4; i += 64) { ((unsigned long long*)P.baseLev)[i] = 0x8000800080008000ull; }          // PAIR_NOCOL everywhere
            for(int i = lane; i < PAIR_COLS / 8; i += 64) { ((unsigned long long*)P.levS)[i] = 0ull; }
            WSYNC();
            int carry = 0;
            for(int t0 = 0; t0 * 64 < nk; t0 += 4) {
                unsigned char ks[4], kg[4]; int kl[4];
                #pragma unroll
                for(int u = 0; u < 4; u++) { const int j = (t0 + u) * 64 + lane; ks[u] = 0; kg[u] = 0; kl[u] = -1; if(j < nk) { ks[u] = B.ext_s[kb + j]; kg[u] = B.ext_g[kb + j]; kl[u] = B.ext_level[kb + j]; } }
                #pragma unroll
                for(int u = 0; u < 4; u++) {
                    if((t0 + u) * 64 >= nk) break;
                    const int j = (t0 + u) * 64 + lane; const bool act = j < nk;
                    const bool isBase = act && ks[u] != '_';
                    const u64 bm = __ballot(isBase);
                    const int rel = kl[u] == -1 ? (int)PAIR_NOLEVEL : kl[u] - firstK;          // (firstK is the first defined level: rel >= 0 for a defined level)
                    if(isBase) { const int bi = carry + __popcll(bm & ((1ull << lane) - 1ull)); if(bi < PAIR_COLS) { P.baseLev[bi] = (short)(rel > 32767 ? 32767 : rel); P.baseG[bi] = kg[u]; } }
                    carry += __popcll(bm);
                    if(act && kl[u] != -1 && firstK >= 0 && rel >= 0 && rel < PAIR_COLS) { P.levS[rel] = ks[u]; P.levG[rel] = kg[u]; }
                }
            }
            WSYNC();
            #pragma unroll
            for(int t = 0; t < PER; t++) {
                const int j = t * 64 + lane;
                if(j >= nSel) continue;
                const int lj = selLv[t]; const unsigned char gj = (unsigned char)(selInfo[t] & 255); const int myIdx = (int)(((u32)selInfo[t] >> 8) & 0xFFFFu);
                bool hit = false;
                if(myIdx != 0xFFFF) {
                    // the other chain's column of the same read base: same level (or both none) and same graph character
                    const short bl = myIdx < PAIR_COLS ? P.baseLev[myIdx] : PAIR_NOCOL;
                    if(bl != PAIR_NOCOL) {
                        int want = lj == -1 ? (int)PAIR_NOLEVEL : lj - firstK;
                        if(lj != -1 && (want > 32767 || want < -32766)) want = 32766;          // out of a chain's span: never stored (stored levels are below PAIR_COLS; a clipped one is 32767)
                        hit = ((int)bl == want) && (P.baseG[myIdx] == gj);
                    }
                }
                else if(lj != -1 && firstK >= 0) { const int li = lj - firstK; if(li >= 0 && li < PAIR_COLS) hit = (P.levS[li] == '_') && (P.levG[li] == gj); }
                if(hit) mask[t] |= (1ull << k);
            }
        }
        WSYNC();
        #pragma unroll
        for(int t = 0; t < PER; t++) {
            const int j = t * 64 + lane;
            if(j >= nSel) continue;
            double Q = 0;                                   // alignmentPositionConfidences accumulated in combination order
            if(m == 0) { for(int i1 = 0; i1 < n1; i1++) if(mask[t] & (1ull << i1)) for(int i2 = 0; i2 < n2; i2++) Q += LL[i1 * n2 + i2]; }
            else       { for(int i1 = 0; i1 < n1; i1++) for(int i2 = 0; i2 < n2; i2++) if(mask[t] & (1ull << i2)) Q += LL[i1 * n2 + i2]; }
            if(Q > 1) Q = 1;
            B.sel_mapq[ob + j] = pair_phred(P, Q);
        }
    }
}

#ifdef HLALA_PAIR_TIMING      // build-time switch: cycles per phase of a wavefront (lists, combinations, maximum + posterior, per-position pass) -> counters[24..31]
#define PAIR_T(i) do { __builtin_amdgcn_s_waitcnt(0); const long long t_ = clock64(); tAcc[i] += t_ - tMark; tMark = t_; } while(0)
#else
#define PAIR_T(i) do { } while(0)
#endif

template <bool UNPAIRED, bool BIG>
__device__ __forceinline__ void pair_finish(const DevGraph& G, const DevTables& T, const DevBatch& B, PairLds& P, double* __restrict__ LL,
                                            const int p, const int n1, const int n2, const int nComb, const int mxc, const PairChain& cA, const PairChain& cB, const int lane, const int stride, long long* tAcc, long long& tMark)
{
    constexpr int NM = UNPAIRED ? 1 : 2;
        // ---- combination log likelihoods, row-major (i1, i2) (:3408-3506)
        if(UNPAIRED) { if(lane < nComb) LL[lane] = cA.ll; }                         // read1_extendedChains_log_likelihoods, :3743 (one chain list: nComb = n1 <= PAIR_CHAINS)
        else {
            // what a combination needs of its two chains -- first / last two levels, strand, log likelihood -- sits in lane k for the k-th chain of either mate (PairChain)
            const int4 fA = cA.fl, fB = cB.fl; const int rA = cA.rev, rB = cB.rev; const double lA = cA.ll, lB = cB.ll;
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
#ifndef PAIR_X_NOINSERT
                if(valid) llIS = (fa0 < fb0) ? pair_insert_ll(G, T, fa2, fa3, fb0, fb1) : pair_insert_ll(G, T, fb2, fb3, fa0, fa1);        // alignerBase.cpp:294, 312
#endif
                double combined = rl64(lA, i1) + rl64(lB, i2);
                combined += llIS;
                if(lane == 0) LL[i] = combined;
            }
        }
        WSYNC();
        PAIR_T(1);
        // ---- first maximum (Utilities::findVectorMax, Utilities.cpp:309-323)
        double mx = -1.0e300; int mi = 0x7FFFFFFF;
        for(int i = lane; i < nComb; i += 64) { double v = LL[i]; if(v > mx) { mx = v; mi = i; } }
        for(int o = 32; o; o >>= 1) { double ov = __shfl_xor(mx, o); int oi = __shfl_xor(mi, o); if(ov > mx || (ov == mx && oi < mi)) { mx = ov; mi = oi; } }
        const int bestI = uni(mi), best1 = bestI / n2, best2 = bestI % n2;
        const int selA = uni(P.L.list[0][best1]), selB = UNPAIRED ? selA : uni(P.L.list[1][best2]);
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
        bool svalid = false;
        if(!UNPAIRED) {          // alignedReadPair_strandsValid of the selected chains (alignerBase.cpp:213-244)
            const int fa0 = __builtin_amdgcn_readlane(cA.fl.x, best1), fa2 = __builtin_amdgcn_readlane(cA.fl.z, best1), fb0 = __builtin_amdgcn_readlane(cB.fl.x, best2), fb2 = __builtin_amdgcn_readlane(cB.fl.z, best2);
            const bool ra = __builtin_amdgcn_readlane(cA.rev, best1) != 0, rb = __builtin_amdgcn_readlane(cB.rev, best2) != 0;
            if(fa0 != -1 && fb0 != -1 && ra != rb) svalid = (!ra) ? (fa0 < fb0) : (fa2 > fb2);
        }
        if(lane == 0) {
            B.pair_status[p] = 0; B.n_comb[p] = nComb; B.pair_ll[p] = mx; B.pair_mapq[p] = mapQ;
            if(UNPAIRED) { B.best_chain[p] = selA; B.mate_mapq[p] = mapQ; B.strands_valid[p] = 0; }                       // forReturn.mapQ = mapQ, :3921
            else {
            B.best_chain[2 * p] = selA; B.best_chain[2 * p + 1] = selB; B.mate_mapq[2 * p] = q1; B.mate_mapq[2 * p + 1] = q2;
            B.strands_valid[p] = svalid ? 1 : 0;
            }
        }
        PAIR_T(2);
#ifndef PAIR_X_NOPOS
        if(!BIG || mxc <= 192) pair_positions<UNPAIRED, 3>(B, P, LL, p, n1, n2, nComb, best1, best2, cA, cB, lane, stride);
        else pair_positions<UNPAIRED, PAIR_COLS / 64>(B, P, LL, p, n1, n2, nComb, best1, best2, cA, cB, lane, stride);
#endif
        PAIR_T(nComb == 1 ? 3 : 4);
#ifdef HLALA_PAIR_TIMING
        tAcc[nComb == 1 ? 5 : 6] += 1;
#endif
}

// ---- lists of extended chains per mate (read1_extendedChains / read2_extendedChains, :3408-3420), error propagation, and the facts of the listed chains in list
// order (PairChain: lane k holds the k-th chain of either mate's list).  cLo / cHi: the chain ranges of the pair's mates (wave-uniform).
// Returns 1 when the pair cannot be scored (a flagged chain, an empty list, more chains / combinations / columns than the tables hold).
template <bool UNPAIRED>
__device__ __forceinline__ int pair_lists(const DevBatch& B, PairListLds& L, const int* cLo, const int* cHi, const int lane, int& n1, int& n2, int& mxcOut, PairChain* cL)
{
    constexpr int NM = UNPAIRED ? 1 : 2;
    int bad = 0;
    int stF[NM], ncF[NM];
    // status and length of the first 64 chains of both mates: one round trip (a pair with more alignments per mate loops on)
    #pragma unroll
    for(int m = 0; m < NM; m++) { const int c = cLo[m] + lane; stF[m] = 1; ncF[m] = 0; if(c < cHi[m]) { stF[m] = B.ext_status[c]; ncF[m] = B.ext_ncols[c]; } }
    // ... and, in the same round trip, everything else the pairing reads of these chains (PairChain; chains that turn out not to be listed cost a few unused loads)
    PairChain cF[NM];
    #pragma unroll
    for(int m = 0; m < NM; m++) { const int c = cLo[m] + lane; cF[m].fl = make_int4(-1, -1, -1, -1); cF[m].rev = 0; cF[m].nk = 0; cF[m].row = 0; cF[m].ll = 0.0; if(c < cHi[m]) { cF[m] = pair_chain_load(B, c); } }
    int mxc = 0;                 // longest listed chain: the per-position pass holds PAIR_COLS columns per chain (long reads come with one alignment each: nComb == 1, any length)
    #pragma unroll
    for(int m = 0; m < NM; m++) {
        const int c0 = cLo[m], c1 = cHi[m];
        int cnt = 0;
        for(int b0 = c0; b0 < c1; b0 += 64) {
            int c = b0 + lane; int st, nc;
            if(b0 == c0) { st = stF[m]; nc = ncF[m]; } else { st = 1; nc = 0; if(c < c1) { st = B.ext_status[c]; nc = B.ext_ncols[c]; } }
            if(__ballot(st < 0)) bad = 1;
            u64 okm = __ballot(st == HLALA_CHAIN_OK);
            if(st == HLALA_CHAIN_OK) { int pos = cnt + __popcll(okm & ((1ull << lane) - 1ull)); if(pos < PAIR_CHAINS) { L.list[m][pos] = c; L.src[m][pos] = (unsigned char)lane; mxc = max(mxc, nc); } }
            cnt += __popcll(okm);
        }
        if(lane == 0) L.nlist[m] = cnt;
        if(cnt < 1 || cnt > PAIR_CHAINS) bad = 1;
    }
    WSYNC();
    n1 = uni(L.nlist[0]); n2 = UNPAIRED ? 1 : uni(L.nlist[1]);
    bad = uni(bad);
    const long long nCombLL = (long long)n1 * n2;
    if(!bad && nCombLL > PAIR_COMB) bad = 1;
    mxc = wave_max_i32(mxc);
    if(!bad && nCombLL > 1 && mxc > PAIR_COLS) bad = 1;
    mxcOut = mxc;
    if(bad) return 1;
    // the listed chains' facts into list order: lane k <- the lane that loaded chain list[k] (a mate with more than 64 alignments: loaded again through the list)
    #pragma unroll
    for(int m = 0; m < NM; m++) {
        const int nl = m ? n2 : n1;
        if(cHi[m] - cLo[m] <= 64) cL[m] = pair_chain_from_lane(cF[m], lane < nl ? (int)L.src[m][lane] : lane);
        else { cL[m].fl = make_int4(-1, -1, -1, -1); cL[m].rev = 0; cL[m].nk = 0; cL[m].row = 0; cL[m].ll = 0.0; if(lane < nl) cL[m] = pair_chain_load(B, L.list[m][lane]); }
    }
    if(UNPAIRED) cL[1] = cL[0];
    return 0;
}

// Round 6: the pairing stage is two kinds of pairs.  74 % of a Graph M batch's pairs have ONE combination: one insert-size term, mapQ 1, a constant per-position
// quality -- 2.6 k cycles, no table.  The others run the posterior and the per-position pass (43 k cycles; LL table, Phred thresholds and the column tables: 6.3 KB of
// LDS, and, compiled into one kernel with the first kind, 63 spilled registers for everybody).  k_pair_chains now finishes the first kind and LISTS the second
// (multiList / work_counter[WC_PAIR_MULTI ...]: class 0 = up to PAIR_COMB_LDS combinations and chains of up to 192 columns, class 1 = the rest), k_pair_multi<., BIG>
// runs a class from its list.  With 0.7 KB of LDS a block of k_pair_chains finds room beside the wide DP class of its batch (seven blocks of 22 KB on a CU).
// UNPAIRED: one read per unit (processBAM::alignOneLongRead :3618-3838 selects the first maximum of the chains' log likelihoods;
// assignMappingQualities_unpaired :3900-4059 is the paired computation with a single, neutral second mate).
// multiBase: first of the four work counters of this pass's lists (count / fetched of class 0, count / fetched of class 1); multiList: [2][n_pairs] pair numbers.
template <bool UNPAIRED>
__global__ __launch_bounds__(64, PAIR_LEAN_WAVES) void k_pair_chains(const DevGraph* __restrict__ Gp, const DevTables* __restrict__ Tp, const DevBatch* __restrict__ Bp,
                                                   const uint8_t* __restrict__ deferPairs, const int deferMode, const int counterIdx,      // deferMode 1: skip deferred pairs, 2: only those
                                                   int* __restrict__ multiList, const int multiBase)
{
    const DevGraph& G = *Gp;
    const DevBatch& B = *Bp;
    __shared__ PairListLds L;
    const int lane = lane_id();
    const DevTables& T = *Tp;
    const int stride = B.stride;

    // pairs are drawn eight at a time (one same-address atomic per pair serialises the grid at the L2)
    constexpr int CHUNK = 8;
    int sweep = 0;
    // PCorrectToPhred(1): the per-position quality of a pair with one combination (Utilities.cpp:178-203; once per block on the table in global memory)
    int ph1 = 0;
    if(lane == 0) { double pWrong = 1e-100; int lo = 0, hi = 255; while(lo < hi) { const int mid = (lo + hi + 1) >> 1; if(pWrong <= T.phred_thr[mid]) lo = mid; else hi = mid - 1; } ph1 = lo; }
    ph1 = __builtin_amdgcn_readfirstlane(ph1);
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
        int cLo[NM], cHi[NM];
        #pragma unroll
        for(int m = 0; m < NM; m++) { cLo[m] = __builtin_amdgcn_readlane(hC[m], hq); cHi[m] = __builtin_amdgcn_readlane(hC[m + 1], hq); }
        int n1 = 0, n2 = 0, mxc = 0; PairChain cL[2];
        const int bad = pair_lists<UNPAIRED>(B, L, cLo, cHi, lane, n1, n2, mxc, cL);
        if(bad) {
            if(lane == 0) { B.pair_status[p] = -1; if(UNPAIRED) B.best_chain[p] = -1; else { B.best_chain[2 * p] = -1; B.best_chain[2 * p + 1] = -1; } B.n_comb[p] = 0; }
        } else if(n1 * n2 > 1) {
            // posterior over several combinations + per-position pass: k_pair_multi, from this pass's list of the pair's class
            if(lane == 0) { const int cls = (n1 * n2 <= PAIR_COMB_LDS && mxc <= 192) ? 0 : 1; const int q = atomicAdd(&B.work_counter[multiBase + 2 * cls], 1); multiList[(size_t)cls * (size_t)B.n_pairs + q] = p; }
        } else {
            // ---- one combination (:3408-3506): its log likelihood is the maximum, the posterior 1 (:4064-4085), every position's confidence 1 (:4155-4311)
            double combined = cL[0].ll;                               // (lane 0 holds the one listed chain of either mate)
            bool svalid = false;
            if(!UNPAIRED) {
                const int fa0 = __builtin_amdgcn_readlane(cL[0].fl.x, 0), fa1 = __builtin_amdgcn_readlane(cL[0].fl.y, 0), fa2 = __builtin_amdgcn_readlane(cL[0].fl.z, 0), fa3 = __builtin_amdgcn_readlane(cL[0].fl.w, 0);
                const int fb0 = __builtin_amdgcn_readlane(cL[1].fl.x, 0), fb1 = __builtin_amdgcn_readlane(cL[1].fl.y, 0), fb2 = __builtin_amdgcn_readlane(cL[1].fl.z, 0), fb3 = __builtin_amdgcn_readlane(cL[1].fl.w, 0);
                const bool ra = __builtin_amdgcn_readlane(cL[0].rev, 0) != 0, rb = __builtin_amdgcn_readlane(cL[1].rev, 0) != 0;
                if(fa0 != -1 && fb0 != -1 && ra != rb) svalid = (!ra) ? (fa0 < fb0) : (fa2 > fb2);          // alignerBase.cpp:213-244
                double llIS = T.is_penalty;
                if(svalid) llIS = (fa0 < fb0) ? pair_insert_ll(G, T, fa2, fa3, fb0, fb1) : pair_insert_ll(G, T, fb2, fb3, fa0, fa1);        // alignerBase.cpp:294, 312
                combined = cL[0].ll + cL[1].ll;
                combined += llIS;
            }
            const int selA = uni(L.list[0][0]), selB = UNPAIRED ? selA : uni(L.list[1][0]);
            if(lane == 0) {
                B.pair_status[p] = 0; B.n_comb[p] = 1; B.pair_ll[p] = combined; B.pair_mapq[p] = 1.0;
                if(UNPAIRED) { B.best_chain[p] = selA; B.mate_mapq[p] = 1.0; B.strands_valid[p] = 0; }
                else { B.best_chain[2 * p] = selA; B.best_chain[2 * p + 1] = selB; B.mate_mapq[2 * p] = 1.0; B.mate_mapq[2 * p + 1] = 1.0; B.strands_valid[p] = svalid ? 1 : 0; }
            }
            #pragma unroll
            for(int m = 0; m < NM; m++) {
                const int nSel = __builtin_amdgcn_readlane(cL[m].nk, 0); const size_t ob = (size_t)(UNPAIRED ? p : 2 * p + m) * stride;
                for(int j = lane; j < nSel; j += 64) B.sel_mapq[ob + j] = (unsigned char)ph1;
            }
        }
        WSYNC();
        }
    }
}

// The pairs with several combinations, from the list k_pair_chains wrote (class BIG ? 1 : 0 of the pass whose counters start at multiBase): lists again (one round
// trip), then combinations, first maximum, posteriors, per-position pass.  BIG = false: up to PAIR_COMB_LDS combinations in LDS and chains of up to 192 columns
// (three columns per lane); BIG = true: the general form (combination table in the wave's HBM scratch when it does not fit, eight columns per lane).
template <bool UNPAIRED, bool BIG>
__global__ __launch_bounds__(64, BIG ? 3 : PAIR_MULTI_WAVES) void k_pair_multi(const DevGraph* __restrict__ Gp, const DevTables* __restrict__ Tp, const DevBatch* __restrict__ Bp,
                                                  const int* __restrict__ multiList, const int multiBase, double* __restrict__ bigLL)                    // bigLL: [gridDim.x][PAIR_COMB]
{
    const DevGraph& G = *Gp;
    const DevBatch& B = *Bp;
    __shared__ PairLds P;
    const int lane = lane_id();
    const DevTables& T = *Tp;
    const int stride = B.stride;
    constexpr int CHUNK = 4;
    constexpr int NM = UNPAIRED ? 1 : 2;
    const int* list = multiList + (BIG ? (size_t)B.n_pairs : (size_t)0);
    const int nList = uni(B.work_counter[multiBase + (BIG ? 2 : 0)]);
    int* fetch = &B.work_counter[multiBase + (BIG ? 3 : 1)];
    if(nList <= 0) return;
    for(int i = lane; i < 256; i += 64) P.phredThr[i] = T.phred_thr[i];
    WSYNC();
    if(lane == 0) P.ired[7] = (int)pair_phred(P, 1.0);
    WSYNC();
    long long tAcc[8] = {0, 0, 0, 0, 0, 0, 0, 0}; long long tMark = clock64();
    for(;;) {
        int q0 = 0;
        if(lane == 0) q0 = atomicAdd(fetch, CHUNK);
        q0 = __builtin_amdgcn_readfirstlane(q0);
        if(q0 >= nList) break;
        const int qEnd = min(q0 + CHUNK, nList);
        // the chunk's pairs and chain ranges: lane q holds entry q0 + q
        int hP = 0, hC[NM + 1];
        #pragma unroll
        for(int m = 0; m <= NM; m++) hC[m] = 0;
        if(lane < CHUNK && q0 + lane < qEnd) {
            hP = list[q0 + lane];
            #pragma unroll
            for(int m = 0; m <= NM; m++) hC[m] = B.chain_off[NM * hP + m];
        }
        for(int q = q0; q < qEnd; q++) {
            const int hq = q - q0;
            const int p = __builtin_amdgcn_readlane(hP, hq);
            int cLo[NM], cHi[NM];
            #pragma unroll
            for(int m = 0; m < NM; m++) { cLo[m] = __builtin_amdgcn_readlane(hC[m], hq); cHi[m] = __builtin_amdgcn_readlane(hC[m + 1], hq); }
            int n1 = 0, n2 = 0, mxc = 0; PairChain cL[2];
            const int bad = pair_lists<UNPAIRED>(B, P.L, cLo, cHi, lane, n1, n2, mxc, cL);
            if(!bad) {                                          // (k_pair_chains listed the pair because its lists were good)
                const int nComb = n1 * n2;
                PAIR_T(0);
                if(!BIG || nComb <= PAIR_COMB_LDS) pair_finish<UNPAIRED, BIG>(G, T, B, P, P.LL, p, n1, n2, nComb, mxc, cL[0], cL[1], lane, stride, tAcc, tMark);
                else pair_finish<UNPAIRED, BIG>(G, T, B, P, bigLL + (size_t)blockIdx.x * PAIR_COMB, p, n1, n2, nComb, mxc, cL[0], cL[1], lane, stride, tAcc, tMark);
            }
            WSYNC();
        }
    }
#ifdef HLALA_PAIR_TIMING
    if(lane == 0) for(int i = 0; i < 7; i++) atomicAdd(&B.counters[24 + i], (u64)tAcc[i]);
#endif
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
                    const size_t so = row_base(B, ch);
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
    const size_t so = (n > 0 ? row_base(B, ch) : (size_t)0), dofs = (size_t)row * stride;
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
        const size_t so = row_base(B, B.best_chain[r]), mo = (size_t)r * B.stride;
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
