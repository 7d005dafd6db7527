// kernel_dp_band.hip -- the extension DP on LINEAR stretches of the graph: anti-diagonals of score cells held in registers, one 16-lane DPP row per call.
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
//   lane = half the diagonal offset: on iteration d lane l of the row holds the cell with i - j = 2 (l - 8) + (d & 1), so the D source is the SAME lane two
//   iterations back and the two gap sources are the same lane and ONE neighbour (row_shr:1 on even, row_shl:1 on odd iterations) of the iteration before;
//   lane order is level order = std::map order (:794-802); keep threshold -16 (:949), X-drop window 15 below the iteration maximum (:1076-1102) and the
//   running-maximum / patience bookkeeping (:1043-1062) take ONE packed row maximum per iteration: (score, first lane in map order);  the order-dependent
//   `diff` rule (:1007-1041) is trivial here -- every real step of a linear stretch without '_' edges changes the score -- so "equal to the running maximum"
//   alone resets the patience;
//   back pointers are 4 bits per cell (GG open / extend, SG open / extend, D from diagonal / GG / SG), 8 iterations per 32-bit word per lane, in LDS;
//   at most one sequence-complete cell per iteration (:982-999), kept per lane; end cell, rand_r tie rule and "x/z" string order as in kernel_dp.hip (:1427-1472).
//
// Four calls per wavefront run in LOCKSTEP (the iteration number is scalar, so the two parities are two straight-line bodies without a branch); a call that
// leaves its band (16 lanes = diagonal offsets -16 .. +15), walks past the levels staged for it or runs longer than BAND_MAXD iterations FAILS OVER to the general
// 16-lane list (k_dp<DpTiny, 0> draws it after its own): results never depend on which kernel ran a call (HLALA_DP_BAND=0 switches this one off; the parity
// suite runs either way).  Outputs are those of k_dp: dp_iters / dp_score / dp_ncols / dp_sb / dp_se / dp_err, the extension columns in the chain's row, the work
// counters (calls, iterations, candidate cells, edges), the linked duplicates of a call (k_dp_items) served from the same iterations.
#include "batch.h"

namespace hlala {

constexpr int BAND_C = 8;                 // lane of diagonal offset 0 (the start cell)
constexpr int BAND_ABSENT = -20000;       // any value below: no cell / no candidate (DP_NEG plus a few gap costs)
constexpr int BAND_WORDS = BAND_MAXD / 8;

struct __align__(16) BandLds {
    u32 bt[BAND_WORDS][16];               // back pointers: 4 bits per (iteration, lane)
    unsigned short steps[BAND_MAXD];      // steps of the chosen path: kind | i << 2 | j << 9
    unsigned char ls[BAND_REACH];         // ls[t] = label of the t-th step away from the start level (lin_label)
    unsigned char rs[BAND_MAXJ];          // rs[t] = t-th read base the call consumes
};

// the predicate of the 16 lanes of this lane's row as a bit mask (every lane of the wave takes part)
__device__ __forceinline__ u32 band_row_bits(bool p) { const u64 b = __ballot(p); return (u32)(b >> (lane_id() & 48)) & 0xFFFFu; }
__device__ __forceinline__ int band_row_sum(int v) { HLALA_ROW_ALLREDUCE(v, op_add_); return v; }

template <bool FWD>
__device__ __forceinline__ void band_pass(const DevGraph& G, const DevBatch& B, const DpItem* __restrict__ items, const u32 rng_seed, const uint8_t* __restrict__ readBases,
                                          const uint8_t* __restrict__ linLabel, const int* __restrict__ linEid, BandLds& S,
                                          u64& accCalls, u64& accIters, u64& accCells, u64& accEdges)
{
    const int lane = lane_id(), g = lane >> 4, gl = lane & 15, rowBase = lane & 48;
    const int rel = gl - BAND_C;
    constexpr int dirPass = FWD ? 1 : 0;
    const int segStart = uni(B.dp_blk[(size_t)(DPL_BAND + dirPass) * B.dp_nblk]);
    const int nItems = uni(B.dp_blk[(size_t)(DPL_BAND + dirPass + 1) * B.dp_nblk]) - segStart;
    const int* srcList = B.dp_list + segStart;
    int* fetchCounter = &B.work_counter[WC_BAND_FETCH + dirPass];
    const int stride = B.stride;
    const int margin = B.dp_band - 1;
    const int levelsL = G.L;
    for(;;) {
        // ---- four items per wavefront: one atomic per draw
        int w0 = 0;
        if(lane == 0) w0 = atomicAdd(fetchCounter, 4);
        w0 = __builtin_amdgcn_readfirstlane(w0);
        if(w0 >= nItems) break;
        const int w = w0 + g;
        const bool has = w < nItems;
        int idx = 0; int4 a = make_int4(-1, 0, 0, 0), b4 = make_int4(0, 0, 0, 0);
        if(has) { idx = srcList[w]; const int4* ip = (const int4*)(items + idx); a = ip[0]; b4 = ip[1]; }
        const int item0 = a.x, rOff = a.y, seqLen = a.z, y0 = a.w, x0 = b4.x;
        const int jmax = has ? (FWD ? seqLen - y0 : y0) : 0;              // read bases the call can consume (k_dp_items: <= BAND_MAXJ)
        int reach = jmax + margin; if(reach > BAND_REACH) reach = BAND_REACH;   // levels staged: every one of them a linear step (k_dp_items: lin_out / lin_in >= reach)
        if(has) {
            for(int t = gl; t < reach; t += 16) S.ls[t] = linLabel[FWD ? x0 + t : x0 - 1 - t];
            for(int t = gl; t < jmax; t += 16) S.rs[t] = readBases[rOff + (FWD ? y0 + t : y0 - 1 - t)];
        }
        WSYNC();

        // ---- the iterations, :531-1105.  D1 / G1 / S1: the lane's cell of the last diagonal after its filter, D2: of the one before.
        int D1 = (has && gl == BAND_C) ? 0 : DP_NEG, G1 = DP_NEG, S1 = DP_NEG, D2 = DP_NEG;       // :495-519
        int curMax = 0, lastInc = 0, firstPos = BAND_C;          // firstPos: (iteration << 4 | lane) of the first cell that carries the running maximum
        int cA = DP_NEG, cB = DP_NEG;                            // D of the lane's sequence-complete cell on its even / odd iteration
        u32 btAcc = 0;
        int cellsAcc = 0, edgesAcc = 0, itersRun = 0;
        bool fail = false, running = has;
        const int diagonals = seqLen + levelsL - 1;              // :431
        int d = 1;
        auto step = [&](auto Pc) {
            constexpr int P = decltype(Pc)::value;               // d & 1
            const bool anyLive = band_row_bits(D1 > BAND_ABSENT || D2 > BAND_ABSENT) != 0;
            if(running) {
                if(d > diagonals || d - lastInc > 40) running = false;                                              // :553 (itersRun stays d - 1)
                else if(!anyLive) { running = false; itersRun = min(lastInc + 40, diagonals); }                     // both frontiers empty: the remaining iterations are no-ops
                else if(d > BAND_MAXD) { running = false; fail = true; }
            }
            const int i = ((d + 1) >> 1) + rel, j = (d >> 1) - rel;
            const int li = min(max(i - 1, 0), BAND_REACH - 1), rj = min(max(j - 1, 0), BAND_MAXJ - 1);
            const int m = (S.ls[li] == S.rs[rj]) ? 2 : -5;                                                         // :582-590
            int Dg, Gg, Ds, Ss;
            if(P == 0) { Dg = D1; Gg = G1; Ds = dpp_mov<0x111>(DP_NEG, D1); Ss = dpp_mov<0x111>(DP_NEG, S1); }     // row_shr:1 -- the cell one level back
            else       { Dg = dpp_mov<0x101>(DP_NEG, D1); Gg = dpp_mov<0x101>(DP_NEG, G1); Ds = D1; Ss = S1; }     // row_shl:1 -- the cell one read base back
            const bool tgtOK = j <= jmax;                                                                         // :576-580, :624-627
            const int cD = D2 + m;
            const int gO = Dg - 6, gE = Gg - 2; const bool gbit = gE > gO; const int GGv = gbit ? gE : gO;         // :621-661, first maximum: open before extend
            const int sO = Ds - 6, sE = Ss - 2; const bool sbit = sE > sO; const int SGv = sbit ? sE : sO;         // :664-754
            int Dv = cD, dsel = 0;                                                                                // :840-865: D candidates, then GG, then SG
            if(GGv > Dv) { Dv = GGv; dsel = 1; }
            if(SGv > Dv) { Dv = SGv; dsel = 2; }
            if(!tgtOK) Dv = DP_NEG;
            const bool exists = Dv > BAND_ABSENT;
            const bool keep = Dv >= -16;                                                                          // :949
            int mk = keep ? (((Dv + 64) << 4) | (FWD ? 15 - gl : gl)) : 0;                                         // (score, first in map order: lowest level)
            HLALA_ROW_ALLREDUCE(mk, op_max_);
            const int mx = (mk >> 4) - 64;
            const bool survive = keep && (mx - Dv) <= 15;                                                         // :1076-1102
            if(running) {
                itersRun = d;
                cellsAcc += exists ? 1 : 0;                                                                       // :492
                edgesAcc += ((D2 > BAND_ABSENT && tgtOK) ? 1 : 0) + ((Ds > BAND_ABSENT) ? 1 : 0);                  // :428, :459
                if(mk != 0) {                                                                                     // :1043-1062
                    if(mx >= curMax) lastInc = d;
                    if(mx > curMax) { curMax = mx; firstPos = (d << 4) | (FWD ? 15 - (mk & 15) : (mk & 15)); }
                }
                if(keep && j == jmax) { if(P == 0) cA = Dv; else cB = Dv; }                                        // :982-999
                if(keep) btAcc |= (u32)((dsel << 2) | (sbit ? 2 : 0) | (gbit ? 1 : 0)) << (4 * ((d - 1) & 7));
                D2 = D1; D1 = survive ? Dv : DP_NEG; G1 = survive ? GGv : DP_NEG; S1 = survive ? SGv : DP_NEG;     // :1104-1105
                if(survive && (gl == 0 || gl == 15 || i >= reach)) fail = true;                                   // the next iteration would step out of the band / past the staged levels
            }
            if((d & 7) == 0) { if(d <= BAND_MAXD) S.bt[(d - 1) >> 3][gl] = btAcc; btAcc = 0; }
            if(fail) running = false;
            d++;
        };
        for(;;) {
            step(std::integral_constant<int, 1>{});
            step(std::integral_constant<int, 0>{});
            if(__ballot(running) == 0) break;
        }
        { const int dl = d - 1; if((dl & 7) != 0 && dl <= BAND_MAXD) S.bt[(dl - 1) >> 3][gl] = btAcc; }
        WSYNC();
        // a survivor on the edge of the band or past the staged levels is seen by one lane: the row fails as a whole
        fail = band_row_bits(fail) != 0;
        const int nCells = band_row_sum(cellsAcc), nEdges = band_row_sum(edgesAcc);

        // ---- end cell, backtrace, columns -- once for the call and once more for every linked duplicate (k_dp_items: same iterations, own random seed)
        int best = max(cA, cB); HLALA_ROW_ALLREDUCE(best, op_max_);
        const bool tA = cA == best && best > BAND_ABSENT, tB = cB == best && best > BAND_ABSENT;
        const int nTies = __popc(band_row_bits(tA)) + __popc(band_row_bits(tB));
        bool live = has && !fail;           // rows still serving an item
        int item = item0;
        int endPos = -1, nSteps = 0;
        bool first = true;
        for(;;) {
            if(__ballot(live) == 0) break;
            // -- end cell, :1381-1517
            int endScore = 0; bool have = false;
            if(first || __ballot(live && nTies > 1) != 0) {
                const int dA = 2 * (jmax + rel);                                                                   // the even iteration on which this lane's cell is sequence-complete (the odd one: dA + 1)
                int pos = -1;
                if(nTies == 1) pos = tA ? ((dA << 4) | gl) : (tB ? (((dA + 1) << 4) | gl) : -1);
                if(__ballot(live && nTies > 1) != 0) {
                    // the tie with exactly `sel` ties before it in the string order of "x/0" (std::set<std::string>, :493, :1431)
                    u32 sd = rng_seed + (u32)item;
                    const int sel = glibc_rand_r(&sd) % (nTies > 0 ? nTies : 1);                                    // Utilities.cpp:922-927
                    const int xA = FWD ? x0 + jmax + 2 * rel : x0 - (jmax + 2 * rel), xB = FWD ? xA + 1 : xA - 1;
                    int rankA = 0, rankB = 0;
                    for(int u = 0; u < 16; u++) {
                        const int ua = __shfl(tA ? 1 : 0, rowBase + u), ub = __shfl(tB ? 1 : 0, rowBase + u);
                        const int uxA = __shfl(xA, rowBase + u), uxB = __shfl(xB, rowBase + u);
                        if(ua) { if(xz_part(uxA, xA) < 0) rankA++; if(xz_part(uxA, xB) < 0) rankB++; }
                        if(ub) { if(xz_part(uxB, xA) < 0) rankA++; if(xz_part(uxB, xB) < 0) rankB++; }
                    }
                    if(nTies > 1) pos = (tA && rankA == sel) ? ((dA << 4) | gl) : ((tB && rankB == sel) ? (((dA + 1) << 4) | gl) : -1);
                }
                HLALA_ROW_ALLREDUCE(pos, op_max_);
                const int newEnd = nTies > 0 ? pos : (curMax > 0 ? firstPos : -1);
                // -- backtrace, :1109-1354: every lane of the row follows the same pointers (LDS broadcasts), lane 0 records the steps
                if(first || newEnd != endPos) {
                    endPos = newEnd;
                    int d_ = endPos >= 0 ? (endPos >> 4) : 0, l_ = endPos & 15, m_ = 0, n = 0, guard = 0;
                    for(;;) {
                        const bool go = live && d_ > 0 && guard < 3 * BAND_MAXD;
                        if(__ballot(go) == 0) break;
                        if(go) {
                            guard++;
                            const int dd = d_ - 1;
                            const int bits = (int)((S.bt[dd >> 3][l_] >> (4 * (dd & 7))) & 15u);
                            const int i_ = ((d_ + 1) >> 1) + l_ - BAND_C, j_ = (d_ >> 1) - (l_ - BAND_C);
                            int kind = -1;
                            if(m_ == 0) { const int ds = bits >> 2; if(ds == 0) { kind = K_DIAG; d_ -= 2; } else m_ = ds; }
                            else if(m_ == 1) { kind = K_GGAP; if(d_ & 1) l_ += 1; d_ -= 1; m_ = (bits & 1) ? 1 : 0; }
                            else { kind = K_SGAP; if(!(d_ & 1)) l_ -= 1; d_ -= 1; m_ = (bits & 2) ? 2 : 0; }
                            if(kind >= 0) { if(gl == 0 && n < BAND_MAXD) S.steps[n] = (unsigned short)(kind | (i_ << 2) | (j_ << 9)); n++; }
                        }
                    }
                    nSteps = n;
                    WSYNC();
                }
            }
            endScore = nTies > 0 ? best : curMax;
            have = endPos >= 0;
            // -- toVerboseSeedChain (VirtualNWUnique.cpp:28-29) and the columns, written into the chain's output row as k_dp does (dp_expand)
            int sb = 0, se = -1, err = 0;
            if(have) {
                const int dE = endPos >> 4, lE = endPos & 15;
                const int jE = (dE >> 1) - (lE - BAND_C);
                if(FWD) { sb = y0; se = y0 + jE - 1; } else { sb = y0 - jE; se = y0 - 1; }
                const int nCols = nSteps;
                if(nCols > stride || nCols > BAND_MAXD) { err = -1000000 - nCols; have = false; }
                else if(sb > se) { err = __LINE__; have = false; }
                else {
                    const int rowOff = FWD ? stride - nCols : sb;
                    if(live && rowOff + nCols <= stride) {
                        const size_t cb = (size_t)(item >> 1) * stride + rowOff;
                        int* oL = B.ext_level + cb; int* oE = B.ext_edge + cb; uint8_t* oG = B.ext_g + cb; uint8_t* oS = B.ext_s + cb;
                        for(int s = gl; s < nCols; s += 16) {
                            const int st = S.steps[s]; const int kind = st & 3, i_ = (st >> 2) & 127, j_ = st >> 9;
                            const int lvl = FWD ? x0 + i_ - 1 : x0 - i_;
                            const int at = FWD ? nCols - 1 - s : s;                        // forward traces are reversed at the end, :1319-1326
                            const unsigned char sc = S.rs[j_ > 0 ? j_ - 1 : 0], gc = S.ls[i_ > 0 ? i_ - 1 : 0];
                            if(kind == K_GGAP) { oL[at] = -1; oE[at] = -1; oG[at] = '_'; oS[at] = sc; }
                            else { oL[at] = lvl; oE[at] = linEid[lvl]; oG[at] = gc; oS[at] = (kind == K_DIAG) ? sc : (unsigned char)'_'; }
                        }
                    }
                }
            }
            int nx = -1;
            if(live && gl == 0) {
                B.dp_iters[item] = itersRun; B.dp_score[item] = have ? endScore : INT32_MIN; B.dp_ncols[item] = have ? nSteps : -1;
                B.dp_sb[item] = sb; B.dp_se[item] = se; B.dp_err[item] = err;
                accCalls++; accIters += (u64)itersRun; accCells += (u64)nCells; accEdges += (u64)nEdges;
                nx = B.dp_alias_head[item]; if(nx < 0) nx = B.dp_alias_next[item];
            }
            nx = __shfl(nx, rowBase);
            if(nx < 0) live = false; else item = nx;
            first = false;
        }
        // ---- a call that left the band goes to the general 16-lane list (with its linked duplicates: that kernel serves them)
        if(has && fail && gl == 0) {
            const int q = atomicAdd(&B.work_counter[WC_FO_COUNT + dirPass], 1);
            B.retry_list[(size_t)(14 + dirPass) * (size_t)B.n_chains + q] = idx;
            atomicAdd(&B.work_counter[WC_BAND_FAILED], 1);
        }
        WSYNC();
    }
}

__global__ __launch_bounds__(64, 5) void k_dp_band(const DevGraph* __restrict__ Gp, const DevBatch* __restrict__ Bp, const DpItem* __restrict__ items, const u32 rng_seed,
                                                   const uint8_t* __restrict__ readBases, const uint8_t* __restrict__ linLabel, const int* __restrict__ linEid)
{
    const DevGraph& G = *Gp;
    const DevBatch& B = *Bp;
    __shared__ BandLds SS[4];
    BandLds& S = SS[lane_id() >> 4];
    u64 accCalls = 0, accIters = 0, accCells = 0, accEdges = 0;          // lane 0 of every row: flushed once
    band_pass<false>(G, B, items, rng_seed, readBases, linLabel, linEid, S, accCalls, accIters, accCells, accEdges);
    band_pass<true>(G, B, items, rng_seed, readBases, linLabel, linEid, S, accCalls, accIters, accCells, accEdges);
    if((lane_id() & 15) == 0 && accCalls) {
        atomicAdd(&B.counters[CNT_DP_CALLS], accCalls); atomicAdd(&B.counters[CNT_DP_ITERS], accIters);
        atomicAdd(&B.counters[CNT_DP_CELLS], accCells); atomicAdd(&B.counters[CNT_EDGES], accEdges);
    }
}

}  // namespace hlala
