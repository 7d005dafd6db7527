// kernel_dp_band2.hip -- the extension DP beside the backbone's GAP STRETCHES: two tracks per level and one gap-path jump, anti-diagonals in registers (round 6).
//
// The calls kernel_dp_band.hip cannot take run on the hashed-frontier machine of kernel_dp.hip; the largest group of them (an eighth of a backbone batch's calls, a third
// of its extension time) sits next to a gap stretch: a base track and a '_' track side by side for 5-25 levels -- or two base tracks until a SNP merges them --, the
// gap path's jump across (Graph::computeGapEdgePaths, Graph/Graph.cpp:347-476; extensionAligner.cpp:757-786).  There the reference's frontier over (level x, read offset
// y, node rank z) is two bands of two tracks, and the whole recurrence (mapper/aligner/extensionAligner.cpp:335-1556) fits the registers of one lane per read offset:
//
//   MAIN band   cell (i, j, z) = (levels walked, read bases consumed, rank) is computed on iteration d = i + j by lane j (as in kernel_dp_band.hip), z = 0, 1 side by side.
//   EARLY band  a jump from (a, j, zA) creates cell (b, j, zB) on iteration a + j + 1, Delta = (b - a) - 1 iterations before the main band gets there, and everything that
//               descends from it is Delta iterations early as well: cell (i, j, z), i >= b, on iteration i + j - Delta, by the SAME lane j in a second set of registers.
//   STEP        the edges between two levels are ONE 64-bit word (FlatGraph::trk_w_*): per (source rank, target rank) pair whether it has a real edge, a '_' edge, which
//               comes first, and which bases label it.  "First maximum in push order" (Utilities.cpp:379-406) over a pair's parallel edges is then a closed form: +2 if
//               some real edge carries the read base, else -5; a sequence gap opens through the first real edge and extends for free through the '_' edge (:664-754).
//   MERGE       when the main band reaches a cell the early band has kept, the reference finds it in `scores` (:951-979): per matrix the strictly greater value
//               overwrites, every overwritten entry resets the patience (:1043-1062).  The early values come back through a ring of the last 32 iterations in LDS.
//   DIFF        "equal to the running maximum" resets the patience only when the step behind the cell's STORED D pointer changes the score (:1007-1041): for a new
//               cell that is a property of the chosen candidate ('_' edges, jumps and free extensions do not); for a cell met again and not improved it reads the
//               CURRENT value of the early pointer's predecessor -- the lanes keep the stored values of the cells of the last two iterations, frontier or not.
//   Back pointers: 11 bits per cell and band, one 64-bit word per lane and iteration in the wavefront's slab in HBM (coalesced); the backtrace follows them.
//
// The algorithm was first written as a CPU model, lane by lane and iteration by iteration, and run beside every DP call of the oracle (tools/band2/band2_model.cpp: 24
// random worlds of the parity sweep and Graph M, every completed call equal in columns, score, iterations, cells and edges); this file is its transliteration.
// A call leaves the class -- fails over to the general 16-lane list, like a call of kernel_dp_band.hip -- when a surviving cell reaches the end of the staged track steps
// (a level with three nodes, a node with five edges, a second jump, a jump of fewer than 4 or more than 29 edges), after MAXD iterations, or with more than 16 tied end
// cells.  Results never depend on which kernel ran a call (HLALA_DP_BAND2=0 switches this one off).
#include "batch.h"

namespace hlala {

constexpr int B2_RING = 32;              // iterations the early band may be ahead of the main band (Delta <= 28)
constexpr int B2_TIES = 16;

template <int GW_> struct Band2Cfg;
template <> struct Band2Cfg<16> { static constexpr int GW = 16, MAXJ = B2_MAXJ16, REACH = 128, MAXD = B2_MAXD; };
template <> struct Band2Cfg<32> { static constexpr int GW = 32, MAXJ = B2_MAXJ32, REACH = 160, MAXD = B2_MAXD; };
template <> struct Band2Cfg<64> { static constexpr int GW = 64, MAXJ = B2_MAXJ64, REACH = 224, MAXD = B2_MAXD; };

template <class C>
struct __align__(16) Band2Lds {
    u64 sw[C::REACH + 1];                 // sw[i]: the word of the step INTO level i (1 .. reach)
    u64 ring[B2_RING][C::GW];             // what the early band left in `scores`: per track D, GG, SG (+ 32; 0 = none) and kept | dsel << 1 | gbit << 4 | ssrc << 5 | sext << 6 | sgap << 7
    u32 steps[C::MAXD];                   // the chosen path: kind | i << 3 | j << 12 | zs << 18 | z << 19 | first column << 20
    int tieX[B2_TIES]; unsigned char tieZ[B2_TIES];
};

enum { B2K_DIAG = 0, B2K_GGAP = 1, B2K_SGAP = 2, B2K_SGAPG = 3, B2K_JUMP = 4 };

__device__ __forceinline__ int b2_base_code(unsigned char c) { return c == 'A' ? 0 : (c == 'C' ? 1 : (c == 'G' ? 2 : (c == 'T' ? 3 : (c == 'N' ? 4 : 5)))); }
__device__ __forceinline__ u32 b2_enc(int v) { return v > -20000 ? (u32)(v + 32) & 255u : 0u; }            // a score as a byte of the ring (scores of kept cells lie in -22 .. 190)
__device__ __forceinline__ int b2_dec(u32 b) { return b ? (int)b - 32 : DP_NEG; }

// One band of one lane: the two cells (ranks 0 and 1) of level i from the cells of the last two iterations (tools/band2/band2_model.cpp, "candidates").
// D2p / D1p / G1p: the lane of read offset j - 1 (cells (i - 1, j - 1) of two iterations ago, (i, j - 1) of the last one); D1o / S1o: this lane's cells (i - 1, j) of
// the last iteration; jv / jz: the jump's candidate and target rank (jz < 0: none).  Per cell: newD, dsel (0 / 1 diagonal from rank 0 / 1, 2 / 3 '_' edge from rank
// 0 / 1, 4 jump, 5 GG, 6 SG), GGv + gbit, SGv + ssel (ssrc | sext << 1 | sgap << 2).
template <bool FWD, bool T1>      // T1 = false: no rank-1 cell and no second track anywhere in the wavefront this iteration -- only the pair (0, 0)
__device__ __forceinline__ void b2_eval(const u64 sw, const bool hasPrev, const int bc, const int* D2p, const int* D1p, const int* G1p, const int* D1o, const int* S1o, const int jv, const int jz,
                                        int* newD, int* dsel, int* GGv, int* gbit, int* SGv, int* ssel, int& edges)
{
    constexpr int ABS = -20000;
    const int deg0 = (int)((sw >> 9) & 7ull), deg1 = (int)((sw >> 41) & 7ull);
    if(sw) {
        if(hasPrev) { if(D2p[0] > ABS) edges += deg0; if(T1 && D2p[1] > ABS) edges += deg1; }       // :428: every edge of a source cell of the m-2 diagonal
        if(D1o[0] > ABS) edges += deg0; if(T1 && D1o[1] > ABS) edges += deg1;                        // :459: ... and of the m-1 diagonal
    }
    constexpr int NZ = T1 ? 2 : 1;
#pragma unroll
    for(int z = 0; z < NZ; z++) {
        const u32 pr[2] = {(u32)((sw >> (16 * z)) & 0xFFFFull), (u32)((sw >> (32 + 16 * z)) & 0xFFFFull)};
        int best = DP_NEG, ds = 0;
        // m-2 diagonal (:565-607): +2 when a real edge of the pair carries the read base, else -5
        if(hasPrev) {
#pragma unroll
            for(int zs = 0; zs < NZ; zs++) {
                const int src = D2p[zs];
                if(src > ABS && (pr[zs] & 1u)) { const int v = src + ((bc < 5 && ((pr[zs] >> (4 + bc)) & 1u)) ? 2 : -5); if(v > best) { best = v; ds = zs; } }
            }
        }
        // m-1 diagonal, D candidates: '_' edges (:738-752) and the jump (:757-786) in map order of their sources -- the jump's source has the lower level: first forward, last backward
        if(FWD) { if(jz == z && jv > ABS && jv > best) { best = jv; ds = 4; } }
#pragma unroll
        for(int zs = 0; zs < NZ; zs++) { const int src = D1o[zs]; if(src > ABS && (pr[zs] & 4u) && src > best) { best = src; ds = 2 + zs; } }
        if(!FWD) { if(jz == z && jv > ABS && jv > best) { best = jv; ds = 4; } }
        // gap in graph (:621-661): open before extend
        int gg = DP_NEG, gb = 0;
        if(hasPrev && D1p[z] > ABS) { gg = D1p[z] - 6; if(G1p[z] > ABS && G1p[z] - 2 > gg) { gg = G1p[z] - 2; gb = 1; } }
        // gap in sequence (:664-754), sources in rank order, per pair: [the '_' edge's free extension if it comes first,] open through the first real edge, extend through it, [the '_' edge's extension]
        int sg = DP_NEG, ss = 0;
#pragma unroll
        for(int zs = 0; zs < NZ; zs++) {
            const int sD = D1o[zs], sS = S1o[zs]; const u32 p = pr[zs];
            if(sD > ABS && (p & 1u)) {
                const bool real = p & 2u, gap = p & 4u, gapFirst = p & 8u;
                const int open = real ? sD - 6 : DP_NEG, extG = (gap && sS > ABS) ? sS : DP_NEG, extR = (real && sS > ABS) ? sS - 2 : DP_NEG;
                if(gapFirst && extG > sg) { sg = extG; ss = zs | 2 | 4; }
                if(open > sg) { sg = open; ss = zs; }
                if(extR > sg) { sg = extR; ss = zs | 2; }
                if(!gapFirst && extG > sg) { sg = extG; ss = zs | 2 | 4; }
            }
        }
        if(gg > best) { best = gg; ds = 5; }                                  // :840-865
        if(sg > best) { best = sg; ds = 6; }
        newD[z] = best; dsel[z] = ds; GGv[z] = gg; gbit[z] = gb; SGv[z] = sg; ssel[z] = ss;
    }
}

// 11-bit back-pointer record of one cell: kept | useD << 1 | useG << 2 | useS << 3 | dsel << 4 | gbit << 7 | ssel << 8
__device__ __forceinline__ u32 b2_rec(bool useD, bool useG, bool useS, int dsel, int gbit, int ssel) { return 1u | (useD ? 2u : 0u) | (useG ? 4u : 0u) | (useS ? 8u : 0u) | ((u32)dsel << 4) | ((u32)gbit << 7) | ((u32)ssel << 8); }

#ifdef HLALA_B2_TIMING       // build-time switch: cycles per phase of a wavefront's task (draw + stage, iterations, end cell + backtrace, columns + outputs) -> counters[16..23]
#define B2_T(i) do { __builtin_amdgcn_s_waitcnt(0); const long long t_ = clock64(); tAcc[i] += t_ - tMark; tMark = t_; } while(0)
#else
#define B2_T(i) do { } while(0)
#endif

template <class C, bool FWD>
__device__ __forceinline__ void band2_pass(const DevGraph& G, const DevBatch& B, const DpItem* __restrict__ items, const u32 rng_seed, const uint8_t* __restrict__ readBases,
                                           Band2Lds<C>& S, u64* __restrict__ slab, u64& accCalls, u64& accIters, u64& accCells, u64& accEdges)
{
    constexpr int GW = C::GW, NG = 64 / GW, ABS = -20000;
    const int lane = lane_id(), g = lane / GW, gl = lane & (GW - 1), rowBase = lane & ~(GW - 1);
    constexpr int dirPass = FWD ? 1 : 0;
    constexpr int listK = (GW == 16 ? DPL_B2_16 : (GW == 32 ? DPL_B2_32 : DPL_B2_64)) + dirPass;
    const int segStart = uni(B.dp_blk[(size_t)listK * B.dp_nblk]);
    const int nItems = uni(B.dp_blk[(size_t)(listK + 1) * B.dp_nblk]) - segStart;
    const int* srcList = B.dp_list + segStart;
    int* fetchCounter = &B.work_counter[WC_B2_FETCH + (GW == 16 ? 0 : (GW == 32 ? 2 : 4)) + dirPass];
    const u64* __restrict__ trkW = FWD ? G.trk_w_out : G.trk_w_in;
    const u32* __restrict__ trkJ = FWD ? G.trk_j_out : G.trk_j_in;
    const int* __restrict__ trkJP = FWD ? G.trk_jp_out : G.trk_jp_in;
    const int stride = B.stride, levelsL = G.L;
#ifdef HLALA_B2_TIMING
    long long tAcc[8] = {0, 0, 0, 0, 0, 0, 0, 0}; long long tMark = clock64();
#endif
    for(;;) {
        B2_T(4);
        int w0 = 0;
        if(lane == 0) w0 = atomicAdd(fetchCounter, NG);
        w0 = __builtin_amdgcn_readfirstlane(w0);
        if(w0 >= nItems) break;
        const int w = w0 + g;
        const bool has = w < nItems;
        int idx = 0; int4 a = make_int4(-1, 0, 0, 0), b4 = make_int4(0, 0, 0, 0);
        if(has) { idx = srcList[w]; const int4* ip = (const int4*)(items + idx); a = ip[0]; b4 = ip[1]; }
        const int item0 = a.x, rOff = a.y, seqLen = a.z, y0 = a.w, x0 = b4.x;
        const int z0 = has ? b4.y - G.level_off[x0] : 0;                  // rank of the start node
        const int trkRun = b4.w;                                          // track steps ahead of the start level (FlatGraph::trk_out / trk_in, capped at 255)
        const int jmax = has ? (FWD ? seqLen - y0 : y0) : 0;              // read bases the call can consume (k_dp_items: <= C::MAXJ)
        // ---- the window: step words and jumps of the levels ahead (tools/band2/band2_model.cpp: build_window)
        int reach = has ? min(trkRun, C::REACH) : 0;
        int t1 = 1 << 20, t2 = 1 << 20; u32 jw1 = 0; int jp1 = -1;        // first / second level (steps from the start) with a jump of more than one edge
        if(has) {
            for(int t = gl; t < reach; t += GW) {
                const int lv = FWD ? x0 + t : x0 - t;
                S.sw[t + 1] = trkW[lv];
                const u32 jw = trkJ[lv];
                if(jw) { if(t < t1) { t2 = t1; t1 = t; jw1 = jw; jp1 = trkJP[lv]; } else if(t < t2) t2 = t; }
            }
        }
        // (per lane t1 < t2 are its two lowest jump levels; the group's: lowest of all t1, then the lowest level above it among all t1 / t2)
        const int gt1 = -grp_max_i32<GW>(-t1);
        const int cand2 = t1 > gt1 ? t1 : t2;
        const int gt2 = -grp_max_i32<GW>(-cand2);
        const int own = (t1 == gt1) ? 1 : 0;
        const u32 gjw = (u32)grp_max_i32<GW>(own ? (int)jw1 : 0); const int gjp = grp_max_i32<GW>(own ? jp1 : -1);
        bool haveJump = false; int ja = 0, jzA = 0, jb = 0, jzB = 0, jlen = 0, jpath = -1;
        {
            int jumpLimit = 1 << 20;
            if(gt1 < (1 << 20)) {
                if(gjw & 1u) { haveJump = true; ja = gt1; jzA = (int)((gjw >> 2) & 1u); jzB = (int)((gjw >> 3) & 1u); jlen = (int)((gjw >> 8) & 255u); jb = ja + jlen; jpath = gjp; jumpLimit = gt2; }
                else jumpLimit = gt1;
            }
            if(jumpLimit < reach) reach = jumpLimit;
            if(haveJump && jb > reach) { if(ja < reach) reach = ja; haveJump = false; }
            if(haveJump && ja >= reach) haveJump = false;
        }
        const int Delta = haveJump ? jlen - 1 : 0;
        const bool wJump = __ballot(haveJump) != 0;                       // (wave-uniform: some call of this wavefront has a jump)
        int myBc = 5;                                                     // the read base this lane's cells consume last, as a code (A C G T N = 0 .. 4)
        if(has && gl >= 1 && gl <= jmax) myBc = b2_base_code(readBases[rOff + (FWD ? y0 + gl - 1 : y0 - gl)]);
        WSYNC();
        B2_T(0);

        // ---- state: [band][rank]; band 0 = main, 1 = early.  D1 / G1 / S1: the lane's cells of the last iteration after the filter, D2: of the one before (frontier values);
        // P*: what `scores` holds for the main band's cells of the last two iterations, frontier or not (the diff rule of cells met again)
        int D1[2][2], G1[2][2], S1[2][2], D2[2][2], PD1[2], PG1[2], PS1[2], PD2[2];
#pragma unroll
        for(int b = 0; b < 2; b++)
#pragma unroll
            for(int z = 0; z < 2; z++) { D1[b][z] = DP_NEG; G1[b][z] = DP_NEG; S1[b][z] = DP_NEG; D2[b][z] = DP_NEG; }
#pragma unroll
        for(int z = 0; z < 2; z++) { PD1[z] = DP_NEG; PG1[z] = DP_NEG; PS1[z] = DP_NEG; PD2[z] = DP_NEG; }
        if(has && gl == 0) { if(z0 == 0) { D1[0][0] = 0; PD1[0] = 0; } else { D1[0][1] = 0; PD1[1] = 0; } }          // :495-519
        int curMax = 0, lastInc = 0, fp = 0;                    // fp: currentMaxima_coordinates.front() as iteration << 8 | lane << 2 | rank << 1 | band
        int cBest = DP_NEG, nTies = 0;                          // lane jmax: best sequence-complete cell, how many equal it (their levels / ranks in S.tieX / tieZ)
        int cellsAcc = 0, edgesAcc = 0, itersRun = 0, dLast = 0;
        int fail = 0; bool running = has && reach >= 1;
        if(has && reach < 1) fail = 2;
        const int diagonals = seqLen + levelsL - 1;              // :431
        const bool inRead = gl <= jmax;
        int d = 1;
        for(;; d++) {
            bool live = false;
#pragma unroll
            for(int b = 0; b < 2; b++)
#pragma unroll
                for(int z = 0; z < 2; z++) live = live || D1[b][z] > ABS || D2[b][z] > ABS;
            const bool anyLive = grp_ballot<GW>(live) != 0;
            if(running) {
                if(d > diagonals || d - lastInc > 40) running = false;                                              // :553 (itersRun stays d - 1)
                else if(!anyLive) { running = false; itersRun = min(lastInc + 40, diagonals); }                     // both frontiers empty: the remaining iterations are no-ops
                else if(d > C::MAXD - 1) { running = false; fail = 4; }
            }
            if(__ballot(running) == 0) break;
            // ---- candidates.  Two wave-uniform switches keep the common iteration short (the loop is bound by instruction issue: ~1 000 vector instructions with everything on):
            // eOn -- some lane of the wavefront holds an early cell or is about to get one through the jump; t1 -- some lane has a rank-1 cell or a step with a second track.
            const int iM = d - gl, iE = d - gl + Delta;
            const u64 swM = (inRead && iM >= 1 && iM <= reach) ? S.sw[iM] : 0ull;
            const bool okM = inRead && iM >= 0 && iM <= reach, okE = inRead && haveJump && iE >= jb && iE <= reach;
            const int jv = (okE && iE == jb) ? (jzA ? D1[0][1] : D1[0][0]) : DP_NEG;       // the jump's candidate: the main band's cell of the last iteration IS the jump's source then (level a, rank zA)
            const bool eOn = wJump && __ballot(haveJump && (D1[1][0] > ABS || D1[1][1] > ABS || D2[1][0] > ABS || D2[1][1] > ABS || jv > ABS)) != 0;
            const u64 swE = (eOn && inRead && haveJump && iE >= max(jb, 1) && iE <= reach) ? S.sw[iE] : 0ull;
            const bool t1 = __ballot(((swM | swE) >> 16) != 0ull || D1[0][1] > ABS || D2[0][1] > ABS || D1[1][1] > ABS || D2[1][1] > ABS || (jv > ABS && jzB == 1)) != 0;
            int nD[2][2], ds[2][2], gv[2][2], gb[2][2], sv[2][2], ss[2][2];
#pragma unroll
            for(int b = 0; b < 2; b++)
#pragma unroll
                for(int z = 0; z < 2; z++) { nD[b][z] = DP_NEG; ds[b][z] = 0; gv[b][z] = DP_NEG; gb[b][z] = 0; sv[b][z] = DP_NEG; ss[b][z] = 0; }
            int edgesIt = 0;
            {
                int D2p[2] = {band_prev<GW>(D2[0][0], gl), DP_NEG}, D1p[2] = {band_prev<GW>(D1[0][0], gl), DP_NEG}, G1p[2] = {band_prev<GW>(G1[0][0], gl), DP_NEG};
                if(t1) { D2p[1] = band_prev<GW>(D2[0][1], gl); D1p[1] = band_prev<GW>(D1[0][1], gl); G1p[1] = band_prev<GW>(G1[0][1], gl);
                         b2_eval<FWD, true>(swM, gl >= 1, myBc, D2p, D1p, G1p, D1[0], S1[0], DP_NEG, -1, nD[0], ds[0], gv[0], gb[0], sv[0], ss[0], edgesIt); }
                else b2_eval<FWD, false>(swM, gl >= 1, myBc, D2p, D1p, G1p, D1[0], S1[0], DP_NEG, -1, nD[0], ds[0], gv[0], gb[0], sv[0], ss[0], edgesIt);
            }
            if(eOn) {
                int D2p[2] = {band_prev<GW>(D2[1][0], gl), DP_NEG}, D1p[2] = {band_prev<GW>(D1[1][0], gl), DP_NEG}, G1p[2] = {band_prev<GW>(G1[1][0], gl), DP_NEG};
                const int jz = (okE && iE == jb) ? jzB : -1;
                if(t1) { D2p[1] = band_prev<GW>(D2[1][1], gl); D1p[1] = band_prev<GW>(D1[1][1], gl); G1p[1] = band_prev<GW>(G1[1][1], gl);
                         b2_eval<FWD, true>(swE, gl >= 1, myBc, D2p, D1p, G1p, D1[1], S1[1], jv, jz, nD[1], ds[1], gv[1], gb[1], sv[1], ss[1], edgesIt); }
                else b2_eval<FWD, false>(swE, gl >= 1, myBc, D2p, D1p, G1p, D1[1], S1[1], jv, jz, nD[1], ds[1], gv[1], gb[1], sv[1], ss[1], edgesIt);
            }
            if(!okM) { nD[0][0] = DP_NEG; nD[0][1] = DP_NEG; }
            if(!okE) { nD[1][0] = DP_NEG; nD[1][1] = DP_NEG; }
            // ---- call maxima (:794-1073)
            // what the early band left in `scores` for the main band's cells of this iteration
            u32 eb[2] = {0, 0};
            if(wJump) {
                const bool ringValid = haveJump && iM >= jb && d - Delta >= 1;
                if(ringValid) { const u64 e = S.ring[(d - Delta) & (B2_RING - 1)][gl]; eb[0] = (u32)e; eb[1] = (u32)(e >> 32); }
            }
            int stD[2][2], stG[2][2], stS[2][2]; bool kept[2][2]; u32 rec[2][2];
            int mxStoredL = DP_NEG, packL = 0; bool anyOv = false, eqNZ = false, needPtr = false;
            bool useDm[2] = {true, true}, useGm[2] = {true, true}, useSm[2] = {true, true};
#pragma unroll
            for(int b = 0; b < 2; b++)
#pragma unroll
                for(int z = 0; z < 2; z++) { kept[b][z] = false; rec[b][z] = 0; stD[b][z] = DP_NEG; stG[b][z] = DP_NEG; stS[b][z] = DP_NEG; }
            // one cell (band b, rank z) through :949-1062 (tools/band2/band2_model.cpp, "call maxima")
#define B2_CELL(b, z) do {                                                                                                                        \
                    const int v = nD[b][z];                                                                                                        \
                    const bool k = v >= -16;                                                                      /* :949 */                       \
                    kept[b][z] = k;                                                                                                                \
                    if(v > ABS) cellsAcc += running ? 1 : 0;                                                      /* :492 */                       \
                    if(k) {                                                                                                                        \
                        int sD = v, sG = gv[b][z], sS = sv[b][z];                                                                                  \
                        bool uD = true, uG = true, uS = true;                                                                                      \
                        const bool E = b == 0 && ((eb[z] >> 24) & 1u);                                                                             \
                        if(E) {                                                                                   /* :951-979 */                   \
                            const int eD = b2_dec(eb[z] & 255u), eG = b2_dec((eb[z] >> 8) & 255u), eS = b2_dec((eb[z] >> 16) & 255u);              \
                            uD = v > eD; uG = sG > eG; uS = sS > eS;                                                                               \
                            if(uD || uG || uS) anyOv = true;                                                                                       \
                            if(!uD) sD = eD; if(!uG) sG = eG; if(!uS) sS = eS;                                                                     \
                            useDm[z] = uD; useGm[z] = uG; useSm[z] = uS;                                                                           \
                        }                                                                                                                          \
                        stD[b][z] = sD; stG[b][z] = sG; stS[b][z] = sS;                                                                            \
                        if(sD > mxStoredL) mxStoredL = sD;                                                                                         \
                        const int i_ = b ? iE : iM;                                                                                                \
                        const int ord = FWD ? ((i_ << 7) | (gl << 1) | z) : (((511 - i_) << 7) | ((63 - gl) << 1) | z);                            \
                        const int pk = ((v + 64) << 16) | (0xFFFF - ord);                                                                          \
                        if(pk > packL) packL = pk;                                                                                                 \
                        if(v == curMax) {                                                                         /* :1007-1041 */                 \
                            if(!E || uD) {                                                                                                         \
                                const int sse = (E && !uS) ? (int)((eb[z] >> 29) & 7u) : ss[b][z];                /* (the SG pointer behind a D that came from SG: the stored one) */ \
                                const bool zero = (ds[b][z] >= 2 && ds[b][z] <= 4) || (ds[b][z] == 6 && (sse & 2) && (sse & 4));                   \
                                if(!zero) eqNZ = true;                                                                                             \
                            } else needPtr = true;                                                                                                 \
                        }                                                                                                                          \
                        rec[b][z] = b2_rec(uD, uG, uS, ds[b][z], gb[b][z], ss[b][z]);                                                              \
                    }                                                                                                                              \
                } while(0)
            B2_CELL(0, 0);
            if(t1) B2_CELL(0, 1);
            if(eOn) { B2_CELL(1, 0); if(t1) B2_CELL(1, 1); }
#undef B2_CELL
            if(wJump && __ballot(needPtr) != 0) {
                // cells met again and not improved in D that equal the running maximum: the early band's pointer, followed through the merged GG / SG pointers of the same cell;
                // the predecessor's CURRENT value (the lanes' P registers: the cells of the last two iterations)
                const int PD2p0 = band_prev<GW>(PD2[0], gl), PD2p1 = band_prev<GW>(PD2[1], gl), PD1p0 = band_prev<GW>(PD1[0], gl), PD1p1 = band_prev<GW>(PD1[1], gl);
                const int PG1p0 = band_prev<GW>(PG1[0], gl), PG1p1 = band_prev<GW>(PG1[1], gl);
#pragma unroll
                for(int z = 0; z < 2; z++) {
                    const bool E = (eb[z] >> 24) & 1u;
                    if(kept[0][z] && E && !useDm[z] && nD[0][z] == curMax) {
                        const int eds = (int)((eb[z] >> 25) & 7u), egb = (int)((eb[z] >> 28) & 1u), ess = (int)((eb[z] >> 29) & 7u);
                        int prev = DP_NEG;
                        if(eds == 0) prev = PD2p0; else if(eds == 1) prev = PD2p1;
                        else if(eds == 2) prev = PD1[0]; else if(eds == 3) prev = PD1[1];
                        else if(eds == 4) prev = b2_dec(eb[z] & 255u);
                        else if(eds == 5) { const int gbm = useGm[z] ? gb[0][z] : egb; prev = gbm ? (z ? PG1p1 : PG1p0) : (z ? PD1p1 : PD1p0); }
                        else { const int sm = useSm[z] ? ss[0][z] : ess; prev = (sm & 2) ? ((sm & 1) ? PS1[1] : PS1[0]) : ((sm & 1) ? PD1[1] : PD1[0]); }
                        if(nD[0][z] - prev != 0) eqNZ = true;
                    }
                }
            }
            const int mk = grp_max_i32<GW>(packL);
            const int mxStored = grp_max_i32<GW>(mxStoredL);
            const bool gOv = wJump && grp_ballot<GW>(anyOv) != 0, gEq = grp_ballot<GW>(eqNZ) != 0;
            if(running) {
                itersRun = d; dLast = d;
                edgesAcc += edgesIt;
                if(gOv) lastInc = d;                                                                                // :1043-1062
                if(mk != 0) {
                    const int mxNew = (mk >> 16) - 64;
                    if(mxNew > curMax) {
                        curMax = mxNew; lastInc = d;
                        const int ord = 0xFFFF - (mk & 0xFFFF);
                        const int fi = FWD ? (ord >> 7) : 511 - (ord >> 7), fj = FWD ? ((ord >> 1) & 63) : 63 - ((ord >> 1) & 63), fz = ord & 1;
                        fp = (d << 8) | (fj << 2) | (fz << 1) | ((fi + fj == d) ? 0 : 1);
                    } else if(gEq) lastInc = d;
                }
            }
            // sequence-complete cells (:982-999): lane jmax
            if(__ballot(running && gl == jmax && (kept[0][0] || kept[0][1] || kept[1][0] || kept[1][1])) != 0) {
                if(running && gl == jmax) {
#pragma unroll
                    for(int b = 0; b < 2; b++)
#pragma unroll
                        for(int z = 0; z < 2; z++) if(kept[b][z]) {
                            const bool wasIn = b == 0 && ((eb[z] >> 24) & 1u);
                            const int sD = stD[b][z];
                            if(!(wasIn && sD == b2_dec(eb[z] & 255u))) {
                                if(sD > cBest) { cBest = sD; nTies = 0; }
                                if(sD == cBest) { if(nTies < B2_TIES) { const int i = b ? iE : iM; S.tieX[nTies] = FWD ? x0 + i : x0 - i; S.tieZ[nTies] = (unsigned char)z; } nTies++; }
                            }
                        }
                }
            }
            if(running) {
                // the main band's lanes remember what `scores` holds for their cells of this iteration; the early band's cells go into the ring
                if(wJump) {
#pragma unroll
                    for(int z = 0; z < 2; z++) {
                        PD2[z] = PD1[z];
                        int pD = DP_NEG, pG = DP_NEG, pS = DP_NEG;
                        if(kept[0][z]) { pD = stD[0][z]; pG = stG[0][z]; pS = stS[0][z]; }
                        else if((eb[z] >> 24) & 1u) { pD = b2_dec(eb[z] & 255u); pG = b2_dec((eb[z] >> 8) & 255u); pS = b2_dec((eb[z] >> 16) & 255u); }
                        PD1[z] = pD; PG1[z] = pG; PS1[z] = pS;
                    }
                    if(haveJump) {
                        u32 e[2];
#pragma unroll
                        for(int z = 0; z < 2; z++) e[z] = kept[1][z] ? (b2_enc(stD[1][z]) | (b2_enc(stG[1][z]) << 8) | (b2_enc(stS[1][z]) << 16) | (1u << 24) | ((u32)ds[1][z] << 25) | ((u32)gb[1][z] << 28) | ((u32)ss[1][z] << 29)) : 0u;
                        S.ring[d & (B2_RING - 1)][gl] = (u64)e[0] | ((u64)e[1] << 32);
                    }
                }
                slab[(size_t)d * 64 + lane] = (u64)(rec[0][0] | (rec[0][1] << 11)) | ((u64)(rec[1][0] | (rec[1][1] << 11)) << 32);
                // filtering (:1076-1102) and the next frontiers (:1104-1105)
#pragma unroll
                for(int b = 0; b < 2; b++)
#pragma unroll
                    for(int z = 0; z < 2; z++) {
                        const bool survive = kept[b][z] && (mxStored - stD[b][z]) <= 15;
                        D2[b][z] = D1[b][z];
                        D1[b][z] = survive ? stD[b][z] : DP_NEG; G1[b][z] = survive ? stG[b][z] : DP_NEG; S1[b][z] = survive ? stS[b][z] : DP_NEG;
                        if(survive && (b ? iE : iM) >= reach) fail = 2;           // the next iteration would walk a step that is not staged
                    }
            }
            // (no fence here: a lane reads only its OWN column of the ring, Delta iterations after it wrote it -- program order --, and a fence would wait for the
            //  iteration's store into the HBM slab: a microsecond per iteration, measured)
            fail = grp_max_i32<GW>(fail);
            if(fail) running = false;
        }
        WSYNC();
        B2_T(1);
#ifdef HLALA_B2_TIMING
        tAcc[5] += d - 1; tAcc[6] += 1;
#endif
        const int nCells = grp_sum_i32<GW>(cellsAcc), nEdges = grp_sum_i32<GW>(edgesAcc);

        // ---- end cell, backtrace, columns -- once for the call and once more for every linked duplicate (k_dp_items: same iterations, own random seed)
        const int srcLane = rowBase + jmax;
        const int best = __shfl(cBest, srcLane), nT = __shfl(nTies, srcLane);
        if(nT > B2_TIES && !fail) fail = 5;
        bool liveG = has && !fail;
        int item = item0;
        int endKey = -1, nSteps = 0, nCols = 0, endJ = 0;
        bool first = true;
        for(;;) {
            if(__ballot(liveG) == 0) break;
            // -- end cell, :1381-1517
            int ex = 0, ez = 0, ej = 0; bool haveEnd = false;
            if(nT >= 1) {
                int pick = 0;
                if(nT > 1) {
                    u32 sd = rng_seed + (u32)item;
                    const int sel = glibc_rand_r(&sd) % nT;                                                        // Utilities.cpp:922-927
                    // the tie with exactly `sel` ties before it in the string order of "x/z" (std::set<std::string>, :493, :1431)
                    const bool mine = gl < nT && gl < B2_TIES;
                    const int myX = mine ? S.tieX[gl] : 0, myZ = mine ? (int)S.tieZ[gl] : 0;
                    int rank = 0;
                    for(int u = 0; u < B2_TIES; u++) {
                        const int ux = __shfl(myX, rowBase + u), uz = __shfl(myZ, rowBase + u);
                        if(u < nT && xz_less(ux, uz, myX, myZ)) rank++;
                    }
                    pick = grp_max_i32<GW>((mine && rank == sel) ? gl : -1);
                    if(pick < 0) pick = 0;
                }
                ex = S.tieX[pick]; ez = (int)S.tieZ[pick]; ej = jmax; haveEnd = true;
            } else if(curMax > 0) {
                const int fd = fp >> 8, fj = (fp >> 2) & 63, fz = (fp >> 1) & 1, fb = fp & 1;
                const int fi = fd - fj + (fb ? Delta : 0);
                ex = FWD ? x0 + fi : x0 - fi; ez = fz; ej = fj; haveEnd = true;
            }
            const int newKey = haveEnd ? ((ex << 8) | (ej << 1) | ez) : -1;
            // -- backtrace, :1109-1354: every lane of the group follows the same pointers (broadcast loads from the slab), its first lane records the steps
            if(liveG && (first || newKey != endKey)) {
                endKey = newKey; endJ = ej;
                int ci = haveEnd ? (FWD ? ex - x0 : x0 - ex) : 0, cj = haveEnd ? ej : 0, cz = ez, cm = 0, n = 0, cols = 0, guard = 0;
                // The pointers live in HBM: a dependent load per step would be a microsecond per step.  Most steps are diagonal -- (iteration - 2, lane - 1) --, so lane q of the
                // group fetches the words of the q-th cell DOWN THE DIAGONAL from an anchor cell in one round trip; a step that stays on the diagonal takes its words from
                // there, one that leaves it (a gap, the jump) moves the anchor.
                int pfT = -1, pfL = -1; u64 pfM = 0, pfE = 0;
                while(!(ci == 0 && cj == 0) && guard < 4 * C::MAXD && !fail) {
                    guard++;
                    const int tm = ci + cj, te = tm - Delta;
                    int q = (pfT >= 0 && ((pfT - tm) & 1) == 0) ? (pfT - tm) >> 1 : -1;
                    if(!(q >= 0 && q < GW && pfL - q == cj)) {
                        pfT = tm; pfL = cj; q = 0;
                        const int t_ = tm - 2 * gl, l_ = cj - gl, te_ = t_ - Delta;
                        pfM = (l_ >= 0 && t_ >= 1 && t_ <= dLast) ? slab[(size_t)t_ * 64 + rowBase + l_] : 0ull;
                        pfE = (haveJump && l_ >= 0 && te_ >= 1 && te_ <= dLast) ? slab[(size_t)te_ * 64 + rowBase + l_] : 0ull;
                    }
                    const u64 wmq = ((u64)(u32)__shfl((int)(pfM >> 32), rowBase + q) << 32) | (u64)(u32)__shfl((int)pfM, rowBase + q);
                    const u64 weq = ((u64)(u32)__shfl((int)(pfE >> 32), rowBase + q) << 32) | (u64)(u32)__shfl((int)pfE, rowBase + q);
                    const u64 wm = (tm >= 1 && tm <= dLast) ? wmq : 0ull;
                    const u64 we = (haveJump && ci >= jb && te >= 1 && te <= dLast) ? weq : 0ull;
                    const u32 rm = (u32)(wm >> (11 * cz)) & 0x7FFu, re = (u32)(we >> (32 + 11 * cz)) & 0x7FFu;
                    const int bit = cm == 0 ? 1 : (cm == 1 ? 2 : 3);
                    const u32 r = ((rm & 1u) && ((rm >> bit) & 1u)) ? rm : ((re & 1u) ? re : rm);
                    if(!(r & 1u)) { fail = 6; break; }                    // (cannot happen: the path only visits kept cells)
                    u32 st = 0; int len = 1;
                    if(cm == 0) {
                        const int dsl = (int)((r >> 4) & 7u);
                        if(dsl <= 1) { st = B2K_DIAG | (ci << 3) | (cj << 12) | (dsl << 18) | (cz << 19); ci -= 1; cj -= 1; cz = dsl; }
                        else if(dsl <= 3) { st = B2K_SGAPG | (ci << 3) | (cj << 12) | ((dsl - 2) << 18) | (cz << 19); ci -= 1; cz = dsl - 2; }
                        else if(dsl == 4) { st = B2K_JUMP | (ci << 3) | (cj << 12); len = jlen; ci = ja; cz = jzA; }
                        else { cm = dsl == 5 ? 1 : 2; continue; }
                    } else if(cm == 1) {
                        st = B2K_GGAP | (ci << 3) | (cj << 12) | (cz << 19); cj -= 1; cm = ((r >> 7) & 1u) ? 1 : 0;
                    } else {
                        const int sse = (int)((r >> 8) & 7u);
                        st = ((sse & 4) ? B2K_SGAPG : B2K_SGAP) | (ci << 3) | (cj << 12) | ((sse & 1) << 18) | (cz << 19); ci -= 1; cz = sse & 1; cm = (sse & 2) ? 2 : 0;
                    }
                    if(n < C::MAXD) { if(gl == 0) S.steps[n] = st | ((u32)cols << 20); } else fail = 7;
                    n++; cols += len;
                }
                if(guard >= 4 * C::MAXD) fail = 8;
                nSteps = n; nCols = cols;
                WSYNC();
            }
            fail = grp_max_i32<GW>(fail);
            if(fail) liveG = false;
            B2_T(2);
            const int endScore = nT >= 1 ? best : curMax;
            bool have = endKey >= 0 && liveG;
            // -- toVerboseSeedChain (VirtualNWUnique.cpp:28-29) and the columns, written into the chain's output row as k_dp does (dp_expand)
            int sb = 0, se = -1, err = 0;
            if(have) {
                if(FWD) { sb = y0; se = y0 + endJ - 1; } else { sb = y0 - endJ; se = y0 - 1; }
                if(nCols > stride) { err = -1000000 - nCols; have = false; }
                else if(sb > se) { err = __LINE__; have = false; }
                else {
                    const int rowOff = FWD ? stride - nCols : sb;
                    if(rowOff + nCols <= stride) {
                        const size_t cb = row_base(B, item >> 1) + rowOff;
                        int* oL = B.ext_level + cb; int* oE = B.ext_edge + cb; uint8_t* oG = B.ext_g + cb; uint8_t* oS = B.ext_s + cb;
                        for(int s = gl; s < nSteps; s += GW) {
                            const u32 st = S.steps[s]; const int kind = st & 7, ci = (st >> 3) & 511, cj = (st >> 12) & 63, zs = (st >> 18) & 1, cz = (st >> 19) & 1, c0 = (int)(st >> 20);
                            const unsigned char sc = cj > 0 ? readBases[rOff + (FWD ? y0 + cj - 1 : y0 - cj)] : (unsigned char)0;
                            if(kind == B2K_JUMP) {                                                                 // :1282-1307
                                const long long po = G.path_off[jpath];
                                const int lvl0 = G.node_level[G.edge_from_new[G.path_edges[po]]];
                                const int at = FWD ? nCols - c0 - jlen : c0;
                                for(int q = 0; q < jlen; q++) { oL[at + q] = lvl0 + q; oE[at + q] = G.path_edges[po + q]; oG[at + q] = '_'; oS[at + q] = '_'; }
                            } else {
                                const int at = FWD ? nCols - 1 - c0 : c0;                                          // forward traces are reversed at the end, :1319-1326
                                if(kind == B2K_GGAP) { oL[at] = -1; oE[at] = -1; oG[at] = '_'; oS[at] = sc; }
                                else {
                                    // the edge of the step: from the source node (rank zs of the level before, walking) to the target node (rank cz), the first one that fits
                                    const int lvl = FWD ? x0 + ci - 1 : x0 - ci;
                                    const int srcNode = G.level_off[FWD ? x0 + ci - 1 : x0 - ci + 1] + zs, tgtNode = G.level_off[FWD ? x0 + ci : x0 - ci] + cz;
                                    const int e0 = (FWD ? G.out_off : G.in_off)[srcNode], e1 = (FWD ? G.out_off : G.in_off)[srcNode + 1];
                                    const int* eto = FWD ? G.out_to : G.in_from; const uint8_t* elab = FWD ? G.out_label : G.in_label;
                                    int pickE = -1, firstE = -1;
                                    for(int q = e0; q < e1; q++) if(eto[q] == tgtNode) {
                                        const unsigned char lb = elab[q];
                                        if(kind == B2K_DIAG) { if(firstE < 0) firstE = q; if(lb == sc && pickE < 0) pickE = q; }
                                        else if((lb == '_') == (kind == B2K_SGAPG)) { if(pickE < 0) pickE = q; }
                                    }
                                    if(pickE < 0) pickE = firstE;
                                    if(pickE < 0) pickE = e0;                     // (cannot happen)
                                    oL[at] = lvl; oE[at] = (FWD ? G.out_eid : G.in_eid)[pickE]; oG[at] = elab[pickE]; oS[at] = kind == B2K_DIAG ? sc : (unsigned char)'_';
                                }
                            }
                        }
                    }
                }
            }
            int nx = -1;
            if(liveG && gl == 0) {
                B.dp_iters[item] = itersRun; B.dp_score[item] = have ? endScore : INT32_MIN; B.dp_ncols[item] = have ? nCols : -1;
                B.dp_sb[item] = sb; B.dp_se[item] = se; B.dp_err[item] = err;
                accCalls++; accIters += (u64)itersRun; accCells += (u64)nCells; accEdges += (u64)nEdges;
                nx = B.dp_alias_head[item]; if(nx < 0) nx = B.dp_alias_next[item];
            }
            nx = __shfl(nx, rowBase);
            if(nx < 0) liveG = false; else item = nx;
            first = false;
            B2_T(3);
        }
        // ---- a call that left the class goes to the general 16-lane list (with its linked duplicates: that kernel serves them)
        if(has && fail && gl == 0) {
            const int q = atomicAdd(&B.work_counter[WC_FO_COUNT + dirPass], 1);
            B.retry_list[(size_t)(14 + dirPass) * (size_t)B.n_chains + q] = idx;
            atomicAdd(&B.work_counter[WC_B2_FAILED], 1); atomicAdd(&B.work_counter[WC_B2_WHY + (fail < 8 ? fail : 7)], 1);
        }
        WSYNC();
    }
#ifdef HLALA_B2_TIMING
    if(lane == 0) { const int base = GW == 16 ? 0 : (GW == 32 ? 8 : 16); for(int i = 0; i < 7; i++) atomicAdd(&B.counters[8 + base + i], (u64)tAcc[i]); }
#endif
}

template <int GW>
__global__ __launch_bounds__(64, 4) void k_dp_band2(const DevGraph* __restrict__ Gp, const DevBatch* __restrict__ Bp, const DpItem* __restrict__ items, const u32 rng_seed,
                                                     const uint8_t* __restrict__ readBases, u64* __restrict__ slabs)
{
    typedef Band2Cfg<GW> C;
    const DevGraph& G = *Gp;
    const DevBatch& B = *Bp;
    __shared__ Band2Lds<C> SS[64 / GW];
    Band2Lds<C>& S = SS[lane_id() / GW];
    u64* slab = slabs + (size_t)blockIdx.x * (size_t)B2_MAXD * 64;
    u64 accCalls = 0, accIters = 0, accCells = 0, accEdges = 0;          // first lane of every group: flushed once
    band2_pass<C, false>(G, B, items, rng_seed, readBases, S, slab, accCalls, accIters, accCells, accEdges);
    band2_pass<C, true>(G, B, items, rng_seed, readBases, S, slab, accCalls, accIters, accCells, accEdges);
    if((lane_id() & (GW - 1)) == 0 && accCalls) {
        atomicAdd(&B.counters[CNT_DP_CALLS], accCalls); atomicAdd(&B.counters[CNT_DP_ITERS], accIters);
        atomicAdd(&B.counters[CNT_DP_CELLS], accCells); atomicAdd(&B.counters[CNT_EDGES], accEdges);
    }
}

}  // namespace hlala
