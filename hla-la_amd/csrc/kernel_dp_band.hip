// kernel_dp_band.hip -- the extension DP on LINEAR stretches of the graph: anti-diagonals of score cells held in registers, one lane per read offset.
//
// extensionAligner::fullNeedleman_diagonal_extension_gapJumper (mapper/aligner/extensionAligner.cpp:335-1556) is a sparse frontier over cells
// (level x, read offset y, node rank z) because the graph branches.  94 % of the levels of an MHC-scale graph hold ONE node, and from most start cells the
// levels a call can reach are a run of such levels joined by single edges (FlatGraph::lin_out / lin_in): there z = 0 everywhere, the frontier is a plain band,
// and the whole machine of kernel_dp.hip -- target hash, compare-and-swap claims, cell table in HBM, early-cell hash, rank sort -- has nothing to do:
//
//   cell (i, j) = (levels walked from the start, read bases consumed); anti-diagonal d = i + j = the iteration that creates it (:531); no cell is met twice;
//   its candidates come from exactly three cells, in the reference's push order: D from (i-1, j-1) of iteration d-2 (:565-607), then GraphGap open / extend
//   from (i, j-1) of d-1 (:621-661), then SequenceGap open / extend from (i-1, j) of d-1 (:664-754); "first maximum in push order" (Utilities.cpp:379-406)
//   is therefore the closed form D > GG > SG, open before extend (:804-865);
//   LANE = READ OFFSET j: on iteration d lane j of a call's group holds cell (d - j, j), so the D and GraphGap sources are lane j - 1 (one DPP shift) of the
//   last two iterations and the SequenceGap source is the lane itself; a lane's read base never changes, its level advances by one per iteration; every cell
//   of a diagonal is in the group whatever its offset -- the chains of sequence gaps that trail the best cells for up to 40 iterations (:553) included;
//   lane order is level order = std::map order (:794-802); keep threshold -16 (:949), X-drop window 15 below the iteration maximum (:1076-1102) and the
//   running-maximum / patience bookkeeping (:1043-1062) take ONE packed group maximum per iteration: (score, first lane in map order);  the order-dependent
//   `diff` rule (:1007-1041) is trivial here -- every real step of a linear stretch without '_' edges changes the score -- so "equal to the running maximum"
//   alone resets the patience;
//   back pointers are 4 bits per cell (GG open / extend, SG open / extend, D from diagonal / GG / SG), 8 iterations per 32-bit word per lane, in LDS;
//   the sequence-complete cells (:982-999) are those of lane `bases left`: one per iteration, best score and ties kept there; end cell, rand_r tie rule and
//   "x/z" string order as in kernel_dp.hip (:1427-1472).
//
// Three instantiations by the read bases a call has left: 16 lanes per call (up to 15 bases, four calls per wavefront), 32 (up to 31, two calls), 64 (up to 48).  The calls
// of a wavefront run in LOCKSTEP (the iteration number is scalar).  A call that walks past the levels staged for it or runs longer than MAXD iterations FAILS OVER to the general
// 16-lane list (k_dp<DpTiny, 0> draws it after its own): results never depend on which kernel ran a call (HLALA_DP_BAND=0 switches this one off; the parity
// suite runs either way).  Outputs are those of k_dp: dp_iters / dp_score / dp_ncols / dp_sb / dp_se / dp_err, the extension columns in the chain's row, the work
// counters (calls, iterations, candidate cells, edges), the linked duplicates of a call (k_dp_items) served from the same iterations.
#include "batch.h"

namespace hlala {

constexpr int BAND_ABSENT = -20000;       // any value below: no cell / no candidate (DP_NEG plus a few gap costs)
constexpr int BAND_DRAW = 16;             // tasks of a wavefront per draw from the item list
constexpr int BAND_TIES = 16;             // equal best sequence-complete cells a call may hold (more: fail-over)

// WAVES: five per SIMD (96 registers, nothing spilled).  Six (80 registers) spilled 8 registers of the 16- and 32-lane instantiations into scratch inside the iteration
// loop and bought nothing: the kernels take the same time with 12, 16 or 24 waves on a CU (profiles/r05_experiments.txt 4).
template <int GW_> struct BandCfg;
template <> struct BandCfg<16> { static constexpr int GW = 16, MAXJ = BAND_MAXJ16, REACH = 64, MAXD = 96, WAVES = 5; };
template <> struct BandCfg<32> { static constexpr int GW = 32, MAXJ = BAND_MAXJ32, REACH = 96, MAXD = 128, WAVES = 5; };
template <> struct BandCfg<64> { static constexpr int GW = 64, MAXJ = BAND_MAXJ64, REACH = 112, MAXD = 160, WAVES = 5; };

template <class C>
struct __align__(16) BandLds {
    u32 bt[C::MAXD / 8][C::GW];           // back pointers: 4 bits per (iteration, lane)
    unsigned short steps[C::MAXD];        // steps of the chosen path: kind | i << 2 | j << 9
    unsigned short tieD[BAND_TIES];       // iterations of the sequence-complete cells that equal the best one (entry 0 lives in a register)
    u32 ls[C::REACH];                     // ls[t] = labels of the (one to four parallel) edges of the t-th step away from the start level (lin_label)
};

// the value lane l - 1 of the call's group holds (DP_NEG for its first lane)
template <int GW> __device__ __forceinline__ int band_prev(int v, int gl)
{
    if(GW == 16) return dpp_mov<0x111>(DP_NEG, v);                                       // row_shr:1
    int r = __builtin_amdgcn_update_dpp(DP_NEG, v, 0x138, 0xF, 0xF, false);              // wave_shr:1
    if(GW == 32) r = (gl == 0) ? DP_NEG : r;
    return r;
}

#ifdef HLALA_BAND_TIMING       // build-time switch: cycles per phase of a wavefront's task (draw + stage, iterations, end cell + backtrace, columns + outputs) -> counters[16..21]
#define BAND_T(i) do { __builtin_amdgcn_s_waitcnt(0); const long long t_ = clock64(); tAcc[i] += t_ - tMark; tMark = t_; } while(0)
#else
#define BAND_T(i) do { } while(0)
#endif

template <class C, bool FWD>
__device__ __forceinline__ void band_pass(const DevGraph& G, const DevBatch& B, const DpItem* __restrict__ items, const u32 rng_seed, const uint8_t* __restrict__ readBases,
                                          const u32* __restrict__ linLabel, const int* __restrict__ linEid, BandLds<C>& S,
                                          u64& accCalls, u64& accIters, u64& accCells, u64& accEdges)
{
    constexpr int GW = C::GW, NG = 64 / GW;
    const int lane = lane_id(), g = lane / GW, gl = lane & (GW - 1), rowBase = lane & ~(GW - 1);
    constexpr int dirPass = FWD ? 1 : 0;
    constexpr int listK = (GW == 16 ? DPL_BAND16 : (GW == 32 ? DPL_BAND32 : DPL_BAND64)) + dirPass;
    const int segStart = uni(B.dp_blk[(size_t)listK * B.dp_nblk]);
    const int nItems = uni(B.dp_blk[(size_t)(listK + 1) * B.dp_nblk]) - segStart;
    const int* srcList = B.dp_list + segStart;
    int* fetchCounter = &B.work_counter[WC_BAND_FETCH + (GW == 16 ? 0 : (GW == 32 ? 2 : 4)) + dirPass];
    const int stride = B.stride;
    const int levelsL = G.L;
#ifdef HLALA_BAND_TIMING
    long long tAcc[6] = {0, 0, 0, 0, 0, 0}; long long tMark = clock64();
#endif
    int chunkNext = 0, chunkEnd = 0;         // (scalar: the items of the current draw)
    for(;;) {
        BAND_T(4);
        // ---- NG items per task of the wavefront, BAND_DRAW tasks per atomic: a draw per task would serialise the grid on the counter's L2 line
        // (one word hands out ~88 draws per microsecond, MI355X_MICROARCH.md: the 86 k tasks of a 262 k-pair batch alone were a millisecond of the kernel's 2.3)
        if(chunkNext >= chunkEnd) {
            int w0 = 0;
            if(lane == 0) w0 = atomicAdd(fetchCounter, NG * BAND_DRAW);
            chunkNext = __builtin_amdgcn_readfirstlane(w0); chunkEnd = chunkNext + NG * BAND_DRAW;
        }
        if(chunkNext >= nItems) break;
        const int w = chunkNext + g;
        chunkNext += NG;
        const bool has = w < nItems;
        int idx = 0; int4 a = make_int4(-1, 0, 0, 0), b4 = make_int4(0, 0, 0, 0);
        if(has) { idx = srcList[w]; const int4* ip = (const int4*)(items + idx); a = ip[0]; b4 = ip[1]; }
        const int item0 = a.x, rOff = a.y, seqLen = a.z, y0 = a.w, x0 = b4.x;
        const int linRun = b4.w;                                          // linear steps ahead of the start level (FlatGraph::lin_out / lin_in, capped at 255)
        const int jmax = has ? (FWD ? seqLen - y0 : y0) : 0;              // read bases the call can consume (k_dp_items: <= C::MAXJ)
        // the labels of every linear step ahead are staged (up to REACH): besides the cells that follow the read, the chains of sequence gaps that leave the
        // best cells -- -6, then -2 per level down to the keep threshold of -16 (:949), for up to 40 iterations (:553) -- walk up to bases + 6 levels further
        const int reach = min(linRun, C::REACH);
        int myRc = 0;                                                     // the read base this lane's cells consume last: base j - 1 of the call
        if(has) {
            for(int t = gl; t < reach; t += GW) S.ls[t] = linLabel[FWD ? x0 + t : x0 - 1 - t];
            if(gl >= 1 && gl <= jmax) myRc = readBases[rOff + (FWD ? y0 + gl - 1 : y0 - gl)];
        }
        const u32 rcRep = (u32)myRc * 0x01010101u;
        WSYNC();
        BAND_T(0);

        // ---- the iterations, :531-1105.  D1 / G1 / S1: the lane's cell of the last diagonal after its filter, D2: of the one before.
        int D1 = (has && gl == 0) ? 0 : DP_NEG, G1 = DP_NEG, S1 = DP_NEG, D2 = DP_NEG;           // :495-519
        int curMax = 0, lastInc = 0, firstPos = 0;               // firstPos: (iteration << 6 | lane) of the first cell that carries the running maximum
        int cBest = DP_NEG, cCount = 0, cFirstD = 0;             // lane jmax: best sequence-complete cell so far, how many equal it, the iteration of the first
        u32 btAcc = 0;
        int cellsAcc = 0, edgesAcc = 0, itersRun = 0;
        int fail = 0; bool running = has;          // fail: why the call goes on to the general class (2 past the staged levels, 3 past the linear run, 4 too many iterations, 5 too many ties)
        const int diagonals = seqLen + levelsL - 1;              // :431
        const bool inRead = gl <= jmax;                          // :576-580, :624-627: no cell beyond the read's end
        int d = 1;
        for(;; d++) {
            const bool anyLive = grp_ballot<GW>(D1 > BAND_ABSENT || D2 > BAND_ABSENT) != 0;
            if(running) {
                if(d > diagonals || d - lastInc > 40) running = false;                                              // :553 (itersRun stays d - 1)
                else if(!anyLive) { running = false; itersRun = min(lastInc + 40, diagonals); }                     // both frontiers empty: the remaining iterations are no-ops
                else if(d > C::MAXD) { running = false; fail = 4; }
            }
            if(__ballot(running) == 0) break;
            const int i = d - gl;
            // the step's parallel edges push their D candidates in CSR order (:565-607): the first maximum is +2 through the first edge whose label is the
            // read base, else -5 through the first edge -- "some byte of the word equals the base" decides the score, the edge is looked up at the backtrace
            const u32 lw = S.ls[min(max(i - 1, 0), C::REACH - 1)], lx = lw ^ rcRep;
            const int m = ((lx - 0x01010101u) & ~lx & 0x80808080u) ? 2 : -5;                                      // :582-590
            const int nPar = 4 - (__clz((int)lw) >> 3);                                                           // edges of the step (0 when it is not staged)
            const int Dd = band_prev<GW>(D2, gl), Dg = band_prev<GW>(D1, gl), Gg = band_prev<GW>(G1, gl);         // the cell one read base back: two / one iteration ago
            const int Ds = D1, Ss = S1;                                                                           // the cell one level back, one iteration ago
            const int cD = Dd + m;
            const int gO = Dg - 6, gE = Gg - 2; const bool gbit = gE > gO; const int GGv = gbit ? gE : gO;         // :621-661, first maximum: open before extend
            const int sO = Ds - 6, sE = Ss - 2; const bool sbit = sE > sO; const int SGv = sbit ? sE : sO;         // :664-754
            int Dv = cD, dsel = 0;                                                                                // :840-865: D candidates, then GG, then SG
            if(GGv > Dv) { Dv = GGv; dsel = 1; }
            if(SGv > Dv) { Dv = SGv; dsel = 2; }
            if(!inRead) Dv = DP_NEG;
            const bool exists = Dv > BAND_ABSENT;
            const bool keep = Dv >= -16;                                                                          // :949
            // (score, first in map order): forward the lowest level is the HIGHEST lane, backward the lowest level (= most levels walked) is the LOWEST lane
            const int mk = grp_max_i32<GW>(keep ? (((Dv + 64) << 6) | (FWD ? gl : GW - 1 - gl)) : 0);
            const int mx = (mk >> 6) - 64;
            const bool survive = keep && (mx - Dv) <= 15;                                                         // :1076-1102
            if(running) {
                itersRun = d;
                cellsAcc += exists ? 1 : 0;                                                                       // :492
                edgesAcc += ((Dd > BAND_ABSENT && inRead) ? nPar : 0) + ((Ds > BAND_ABSENT) ? nPar : 0);           // :428, :459: every edge of a source cell counts
                if(mk != 0) {                                                                                     // :1043-1062
                    if(mx >= curMax) lastInc = d;
                    if(mx > curMax) { curMax = mx; firstPos = (d << 6) | (FWD ? (mk & 63) : GW - 1 - (mk & 63)); }
                }
                if(keep && gl == jmax) {                                                                          // :982-999
                    if(Dv > cBest) { cBest = Dv; cCount = 1; cFirstD = d; }
                    else if(Dv == cBest) { if(cCount < BAND_TIES) S.tieD[cCount] = (unsigned short)d; cCount++; }
                }
                if(keep) btAcc |= (u32)((dsel << 2) | (sbit ? 2 : 0) | (gbit ? 1 : 0)) << (4 * ((d - 1) & 7));
                D2 = D1; D1 = survive ? Dv : DP_NEG; G1 = survive ? GGv : DP_NEG; S1 = survive ? SGv : DP_NEG;     // :1104-1105
                if(survive && i >= reach) fail = (reach < linRun) ? 2 : 3;          // the next iteration would walk a level that is not staged / not linear
            }
            // back pointers of the last eight iterations into their word (every call of the wavefront, running or not: a call that stopped keeps its last bits until here)
            if((d & 7) == 0) { S.bt[(d - 1) >> 3][gl] = btAcc; btAcc = 0; }
            if(fail) running = false;
        }
        // (the loop left at iteration d without computing it: iterations 1 .. d - 1 ran, the words up to the last multiple of 8 are written)
        if(((d - 1) & 7) != 0) S.bt[(d - 1) >> 3][gl] = btAcc;
        WSYNC();
        BAND_T(1);
#ifdef HLALA_BAND_TIMING
        tAcc[5] += d - 1;
#endif
        // a level past the staged ones is seen by one lane: the call fails as a whole
        fail = grp_max_i32<GW>(fail);
        const int nCells = grp_sum_i32<GW>(cellsAcc), nEdges = grp_sum_i32<GW>(edgesAcc);

        // ---- end cell, backtrace, columns -- once for the call and once more for every linked duplicate (k_dp_items: same iterations, own random seed)
        const int srcLane = rowBase + jmax;
        const int best = __shfl(cBest, srcLane), nTies = __shfl(cCount, srcLane), tie0 = __shfl(cFirstD, srcLane);
        if(nTies > BAND_TIES) fail = 5;
        bool live = has && !fail;           // groups still serving an item
        if(live && gl == 0 && nTies > 1) atomicAdd(&B.work_counter[WC_BAND_TIED], 1);       // (statistics: calls whose end cell was drawn among equal ones -- the tests want that path taken)
        int item = item0;
        int endPos = -1, nSteps = 0;
        bool first = true;
        for(;;) {
            if(__ballot(live) == 0) break;
            // -- end cell, :1381-1517
            if(first || __ballot(live && nTies > 1) != 0) {
                int pos = -1;
                if(nTies == 1) pos = (tie0 << 6) | jmax;
                if(__ballot(live && nTies > 1) != 0) {
                    // the tie with exactly `sel` ties before it in the string order of "x/0" (std::set<std::string>, :493, :1431)
                    u32 sd = rng_seed + (u32)item;
                    const int sel = glibc_rand_r(&sd) % (nTies > 0 ? nTies : 1);                                    // Utilities.cpp:922-927
                    const bool mine = nTies > 1 && gl < nTies && gl < BAND_TIES;
                    const int myD = mine ? (gl == 0 ? tie0 : (int)S.tieD[gl]) : 0;
                    const int myX = FWD ? x0 + (myD - jmax) : x0 - (myD - jmax);
                    int rank = 0;
                    for(int u = 0; u < BAND_TIES; u++) {
                        const int ux = __shfl(myX, rowBase + u);
                        if(u < nTies && xz_part(ux, myX) < 0) rank++;
                    }
                    if(nTies > 1) pos = (mine && rank == sel) ? ((myD << 6) | jmax) : -1;
                }
                pos = grp_max_i32<GW>(pos);
                const int newEnd = nTies > 0 ? pos : (curMax > 0 ? firstPos : -1);
                // -- backtrace, :1109-1354: every lane of the group follows the same pointers (LDS broadcasts), its first lane records the steps
                if(first || newEnd != endPos) {
                    endPos = newEnd;
                    int d_ = endPos >= 0 ? (endPos >> 6) : 0, l_ = endPos & 63, m_ = 0, n = 0, guard = 0;
                    for(;;) {
                        const bool go = live && d_ > 0 && guard < 3 * C::MAXD;
                        if(__ballot(go) == 0) break;
                        if(go) {
                            guard++;
                            const int dd = d_ - 1;
                            const int bits = (int)((S.bt[dd >> 3][l_ & (GW - 1)] >> (4 * (dd & 7))) & 15u);
                            const int i_ = d_ - l_, j_ = l_;
                            int kind = -1;
                            if(m_ == 0) { const int ds = bits >> 2; if(ds == 0) { kind = K_DIAG; d_ -= 2; l_ -= 1; } else m_ = ds; }
                            else if(m_ == 1) { kind = K_GGAP; d_ -= 1; l_ -= 1; m_ = (bits & 1) ? 1 : 0; }
                            else { kind = K_SGAP; d_ -= 1; m_ = (bits & 2) ? 2 : 0; }
                            if(kind >= 0) { if(gl == 0 && n < C::MAXD) S.steps[n] = (unsigned short)(kind | (i_ << 2) | (j_ << 9)); n++; }
                        }
                    }
                    nSteps = n;
                    WSYNC();
                }
            }
            BAND_T(2);
            const int endScore = nTies > 0 ? best : curMax;
            bool have = endPos >= 0;
            // -- toVerboseSeedChain (VirtualNWUnique.cpp:28-29) and the columns, written into the chain's output row as k_dp does (dp_expand)
            int sb = 0, se = -1, err = 0;
            if(have) {
                const int jE = endPos & 63;
                if(FWD) { sb = y0; se = y0 + jE - 1; } else { sb = y0 - jE; se = y0 - 1; }
                const int nCols = nSteps;
                if(nCols > stride || nCols > C::MAXD) { err = -1000000 - nCols; have = false; }
                else if(sb > se) { err = __LINE__; have = false; }
                else {
                    const int rowOff = FWD ? stride - nCols : sb;
#ifdef BAND_X_NOCOLS
                    if(false) {
#else
                    if(live && rowOff + nCols <= stride) {
#endif
                        const size_t cb = row_base(B, item >> 1) + rowOff;
                        int* oL = B.ext_level + cb; int* oE = B.ext_edge + cb; uint8_t* oG = B.ext_g + cb; uint8_t* oS = B.ext_s + cb;
                        for(int s = gl; s < nCols; s += GW) {
                            const int st = S.steps[s]; const int kind = st & 3, i_ = (st >> 2) & 127, j_ = st >> 9;
                            const int lvl = FWD ? x0 + i_ - 1 : x0 - i_;
                            const int at = FWD ? nCols - 1 - s : s;                        // forward traces are reversed at the end, :1319-1326
                            const unsigned char sc = j_ > 0 ? readBases[rOff + (FWD ? y0 + j_ - 1 : y0 - j_)] : (unsigned char)0;
                            const u32 lw = S.ls[i_ > 0 ? i_ - 1 : 0];
                            // the edge of the step: a diagonal step took the first edge that carries the read base (else the first), a sequence gap the first --
                            // its open / extend candidates are equal over the parallel edges and the first of equal candidates wins (:664-754, Utilities.cpp:379-406)
                            int par = 0;
                            if(kind == K_DIAG) { if(((lw >> 8) & 255u) == sc) par = 1; if(((lw >> 16) & 255u) == sc && par == 0) par = 2; if((lw >> 24) == sc && par == 0) par = 3; if((lw & 255u) == sc) par = 0; }
                            if(kind == K_GGAP) { oL[at] = -1; oE[at] = -1; oG[at] = '_'; oS[at] = sc; }
                            else {
                                oL[at] = lvl; oE[at] = par == 0 ? linEid[lvl] : G.out_eid[G.out_off[G.level_off[lvl]] + par];
                                oG[at] = (unsigned char)((lw >> (8 * par)) & 255u); oS[at] = (kind == K_DIAG) ? sc : (unsigned char)'_';
                            }
                        }
                    }
                }
            }
            int nx = -1;
            if(live && gl == 0) {
                B.dp_iters[item] = itersRun; B.dp_score[item] = have ? endScore : INT32_MIN; B.dp_ncols[item] = have ? nSteps : -1;
                B.dp_sb[item] = sb; B.dp_se[item] = se; B.dp_err[item] = err;
                accCalls++; accIters += (u64)itersRun; accCells += (u64)nCells; accEdges += (u64)nEdges;
#ifndef BAND_X_NOALIAS
                nx = B.dp_alias_head[item]; if(nx < 0) nx = B.dp_alias_next[item];
#endif
            }
            nx = __shfl(nx, rowBase);
            if(nx < 0) live = false; else item = nx;
            first = false;
            BAND_T(3);
        }
        // ---- a call that left its staged levels goes to the general 16-lane list (with its linked duplicates: that kernel serves them)
        if(has && fail && gl == 0) {
            const int q = atomicAdd(&B.work_counter[WC_FO_COUNT + dirPass], 1);
            B.retry_list[(size_t)(14 + dirPass) * (size_t)B.n_chains + q] = idx;
            atomicAdd(&B.work_counter[WC_BAND_FAILED], 1); atomicAdd(&B.work_counter[WC_BAND_WHY + fail], 1);
        }
        WSYNC();
#ifdef HLALA_BAND_TIMING
        tAcc[4] += 0; if(lane == 0) atomicAdd(&B.counters[22], 1ull);
#endif
    }
#ifdef HLALA_BAND_TIMING
    if(lane == 0 && GW == 16) { for(int i = 0; i < 5; i++) atomicAdd(&B.counters[16 + i], (u64)tAcc[i]); atomicAdd(&B.counters[21], (u64)tAcc[5]); }
#endif
}

template <int GW>
__global__ __launch_bounds__(64, BandCfg<GW>::WAVES) void k_dp_band(const DevGraph* __restrict__ Gp, const DevBatch* __restrict__ Bp, const DpItem* __restrict__ items, const u32 rng_seed,
                                                                    const uint8_t* __restrict__ readBases, const u32* __restrict__ linLabel, const int* __restrict__ linEid)
{
    typedef BandCfg<GW> C;
    const DevGraph& G = *Gp;
    const DevBatch& B = *Bp;
    __shared__ BandLds<C> SS[64 / GW];
    BandLds<C>& S = SS[lane_id() / GW];
    u64 accCalls = 0, accIters = 0, accCells = 0, accEdges = 0;          // first lane of every group: flushed once
    band_pass<C, false>(G, B, items, rng_seed, readBases, linLabel, linEid, S, accCalls, accIters, accCells, accEdges);
    band_pass<C, true>(G, B, items, rng_seed, readBases, linLabel, linEid, S, accCalls, accIters, accCells, accEdges);
    if((lane_id() & (GW - 1)) == 0 && accCalls) {
        atomicAdd(&B.counters[CNT_DP_CALLS], accCalls); atomicAdd(&B.counters[CNT_DP_ITERS], accIters);
        atomicAdd(&B.counters[CNT_DP_CELLS], accCells); atomicAdd(&B.counters[CNT_EDGES], accEdges);
    }
}

}  // namespace hlala
