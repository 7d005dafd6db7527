// kernel_dp_lane.hip -- the lane-per-DP class of the extension stage: 64 DP calls per wavefront, one per LANE.
//
// STATUS: an experiment that LOST, kept switchable (HLALA_DP_LANE=1 at hlala_create) and under test (tests/test_gpu_align.py).  Bit-exact against the
// oracle and the other classes, it finishes 68 % of the DP calls of the Graph M workload (3.77 M calls per million pairs; 1.19 M go on to the 16-lane class),
// but takes 69-78 ms for them where the 16-lane class takes 33 ms (90 ms for all calls, 57 ms for the 1.19 M that are left): a lane walks the candidates
// of its few cells one after the other, the wavefront executes the longest lane's loops, and at one wave per SIMD (36 KB of LDS per wave) every vector
// instruction costs its two cycles of the SIMD-32: ~9 k wave instructions per trip for 64 calls = 140 per call and iteration, against 357 / 4 = 89 in the 16-lane class,
// whose 16 lanes share the work of a call's cells.  Requesting the node records of four frontier entries at a time and writing the columns during the
// backtrace did not change that (the class is bound by issue, not by its chain of loads).  See DESIGN.md, section 4B.
//
// Most calls of the frontier DP (extensionAligner::fullNeedleman_diagonal_extension_gapJumper, mapper/aligner/extensionAligner.cpp:335-1556) are small:
// on 2x150 bp pairs three quarters of them never hold more than a handful of frontier cells, walk nodes with at most two edges and meet no gap-path jump.
// Run by a GROUP of 16 lanes (k_dp<DpTiny>) such a call keeps three or four lanes busy and pays for the group's collectives, the shared target hash and
// the state machine of four groups per wave: ~60 wave-level vector instructions per evaluated cell.  Here every lane runs its own call, serially over its
// few cells: no collectives at all (running maximum, patience, slot numbers and the frontier order are plain per-lane arithmetic), the candidates of an
// iteration go into a private 16-entry table in LDS, the frontier lives in a private LDS column, cells go to a private slab in HBM exactly as in the group
// classes (same 32-byte records, same back pointers), and 64 backtraces chase their pointers side by side.
//
// The class only takes what it can finish EXACTLY as the other classes would: a call leaves for the 16-lane class (it is queued on that class's list and
// leaves no trace here) as soon as it meets a node with more than two edges or with a gap-path jump (-> cells could be reached early and merged,
// extensionAligner.cpp:951-979), more than LN_FC frontier cells, more than 12 targets in one iteration, more than LN_CELLS kept cells, LN_COMPLETED
// sequence-complete cells or LN_STEPS backtrace steps.  Without jumps every target of iteration d lies on its natural diagonal, so no cell is ever met twice:
// no early-cell hash, no staged improvements, the `diff` rule (:1007-1041) reads the cached frontier scores.
// (compiled after kernel_dp.hip in the unity build of hlala_api.hip: its keys, back pointers, candidate packing and helpers are used here)

namespace hlala {

constexpr int LN_FC = 8;             // frontier cells per call
constexpr int LN_TC = 16;            // entries of the per-call target table (at most 12 targets per iteration)
constexpr int LN_TMAX = 12;
constexpr int LN_CELLS = 1024;       // kept cells per call
constexpr int LN_COMPLETED = 32;     // sequence-complete cells per call
constexpr int LN_STEPS = 512;        // backtrace steps per call
struct DpLane { static constexpr int IBITS = 3; typedef u32 Best; };      // push index: phase (1) | frontier entry (3) | rank (8), packed with the score as in the group classes

// scratch of one lane in HBM
struct LaneSlab {
    static constexpr size_t O_CELL = 0;                                            // CellRec[LN_CELLS]
    static constexpr size_t O_STEP_KEY = O_CELL + (size_t)LN_CELLS * 32;           // u64[LN_STEPS]: the cell a step arrives at
    static constexpr size_t O_STEP_BT = O_STEP_KEY + (size_t)LN_STEPS * 8;         // u32[LN_STEPS]
    static constexpr size_t O_COMPLETED = O_STEP_BT + (size_t)LN_STEPS * 4;        // int[LN_COMPLETED]
    static constexpr size_t BYTES = (O_COMPLETED + (size_t)LN_COMPLETED * 4 + 255) & ~(size_t)255;
};
__host__ __device__ inline size_t dp_lane_slab_bytes() { return LaneSlab::BYTES; }

// per-wave LDS: column `lane` of every array belongs to that lane's call (a lane only ever touches its own column: no synchronisation)
struct __align__(16) LaneLds {
    u32 tkLo[LN_TC][64], tkHi[LN_TC][64];          // target cell keys (all ones = free)
    u32 tb[3][LN_TC][64];                          // best candidate per matrix (pack_best)
    u32 fkLo[2][LN_FC][64], fkHi[2][LN_FC][64];    // frontier cells: [0] the last diagonal, [1] the one before (sorted by key = std::map order)
    u32 fA[2][LN_FC][64];                          // table slot (16) | D score (16, as short)
    u32 fB[2][LN_FC][64];                          // GG score (16) | SG score (16)
};

enum { LPH_IDLE = 0, LPH_RUN, LPH_FINISH, LPH_LEAVE };

// work_counter (batch.h): [40]/[42] items left / right that go on to the 16-lane class, [41]/[43] fetched by that class; [44]/[45] fetched here
constexpr int WC_TINY_COUNT = 40, WC_TINY_FETCH = 41, WC_LANE_FETCH = 44;

__global__ __launch_bounds__(64, 1) void k_dp_lane(const DevGraph* __restrict__ Gp, const DevBatch* __restrict__ Bp, const DpItem* __restrict__ items,
                                                   char* slabs, u32 rng_seed, const int4* __restrict__ nrecOut, const int4* __restrict__ nrecIn,
                                                   const uint8_t* __restrict__ readBases, int* __restrict__ tinyList)
{
    const DevGraph& G = *Gp;
    const DevBatch& B = *Bp;
    const int lane = lane_id();
    __shared__ LaneLds S;
    __shared__ int outBuf[128];
    char* const slab = slabs + ((size_t)blockIdx.x * 64 + (size_t)lane) * LaneSlab::BYTES;
    CellRec* const cells = (CellRec*)(slab + LaneSlab::O_CELL);
    u64* const stepKey = (u64*)(slab + LaneSlab::O_STEP_KEY);
    u32* const stepBt = (u32*)(slab + LaneSlab::O_STEP_BT);
    int* const completed = (int*)(slab + LaneSlab::O_COMPLETED);
    u64 accCalls = 0, accIters = 0, accCells = 0, accEdges = 0;

    for(int dirPass = 0; dirPass < 2; dirPass++) {
        const bool fwd = dirPass != 0;
        const int dir = fwd ? 1 : -1;
        const int nItems = ordered_chains(B);           // slots of k_dp_items' lists (one per chain in position order; item = -1: no DP)
        const int listBase = dirPass ? B.n_chains : 0;
        int* const fetchCounter = &B.work_counter[WC_LANE_FETCH + dirPass];
        int* const tinyCount = &B.work_counter[WC_TINY_COUNT + 2 * dirPass];
        int* const tinyOut = tinyList + (size_t)dirPass * (size_t)B.n_chains;
        const int4* const nrec = fwd ? nrecOut : nrecIn;
        const int max_levelI = G.L - 1;
        // ---- the call of this lane
        int phase = LPH_IDLE; bool more = true;
        // items are drawn from the batch's counter 128 at a time and handed to idle lanes from this wave's share; items that leave are collected in LDS
        // and queued 64 at a time (one same-address atomic per item would serialise the grid at the L2)
        int poolNext = 0, poolEnd = 0, nOut = 0;
        int itemIdx = 0, item = 0, rOff = 0, seqLen = 0, start_seq = 0, startLevel = 0, startNode = 0, diagonals = 0;
        int d = 0, n1 = 0, n2 = 0, nCells = 0, nCompleted = 0, curMax = 0, firstMaxSlot = 0, lastInc = 0, itersRun = 0, edges = 0;
        u32 cellsEvaluated = 0;

        for(;;) {
            // ---- idle lanes draw the next items (one atomic per wave and round)
            {
                const u64 want = __ballot(phase == LPH_IDLE && more);
                if(want) {
                    const int nWant = (int)__popcll(want);
                    // (wave-uniform) an empty share is refilled; lanes beyond what is left of a share draw in the next round
                    if(poolNext >= poolEnd) { int nb = 0; if(lane == 0) nb = atomicAdd(fetchCounter, 128); nb = __builtin_amdgcn_readfirstlane(nb); poolNext = nb; poolEnd = nb + 128; }
                    const int rk = (int)__popcll(want & ((1ull << lane) - 1ull));
                    const int avail = poolEnd - poolNext;
                    if(phase == LPH_IDLE && more && rk < avail) {
                        const int w = poolNext + rk;
                        const int4* ip = (const int4*)(items + listBase + (w < nItems ? w : 0));
                        const int4 a = ip[0];
                        if(w >= nItems) more = false;
                        else if(a.x >= 0) {            // (an empty slot: the lane stays idle and draws again in the next round)
                            itemIdx = listBase + w;
                            const int4 b = ip[1];
                            item = a.x; rOff = a.y; seqLen = a.z; start_seq = a.w; startLevel = b.x; startNode = b.y;
                            diagonals = seqLen + G.L - 1;
                            d = 1; n1 = 1; n2 = 0; nCells = 1; nCompleted = 0; curMax = 0; firstMaxSlot = 0; lastInc = 0; itersRun = 0; edges = 0; cellsEvaluated = 0;
                            const u64 k0 = mk_key(startLevel, start_seq, startNode);
                            uint4* dst = (uint4*)cells;
                            dst[0] = make_uint4((u32)k0, (u32)(k0 >> 32), ((u32)(unsigned short)(short)0) | ((u32)(unsigned short)(short)DP_NEG << 16), (u32)(unsigned short)(short)DP_NEG);
                            dst[1] = make_uint4(0u, 0u, 0u, 0u);
                            S.fkLo[0][0][lane] = (u32)k0; S.fkHi[0][0][lane] = (u32)(k0 >> 32);
                            S.fA[0][0][lane] = 0u | ((u32)(unsigned short)(short)0 << 16);
                            S.fB[0][0][lane] = ((u32)(unsigned short)(short)DP_NEG) | ((u32)(unsigned short)(short)DP_NEG << 16);
#pragma unroll
                            for(int t = 0; t < LN_TC; t++) { S.tkLo[t][lane] = 0xFFFFFFFFu; S.tkHi[t][lane] = 0xFFFFFFFFu; S.tb[0][t][lane] = 0; S.tb[1][t][lane] = 0; S.tb[2][t][lane] = 0; }
                            phase = LPH_RUN;
                        }
                    }
                    poolNext += nWant < avail ? nWant : avail;
                }
            }
            if(!__ballot(phase != LPH_IDLE)) break;

            // ================= one iteration (extensionAligner.cpp:531-1105) for the lanes that run =================
            if(phase == LPH_RUN) {
                const uint8_t* seqp = readBases + rOff;
                const int limitY = fwd ? seqLen : 0;
                if(d > diagonals || (d - lastInc) > 40) phase = LPH_FINISH;                                     // :553 maximum_steps_nonIncrease
                else if(n1 == 0 && n2 == 0) { int last = lastInc + 40; if(last > diagonals) last = diagonals; itersRun = last; phase = LPH_FINISH; }
                else if(d > 60000) phase = LPH_LEAVE;
                else {
                    bool leave = false;
                    int nT = 0;
                    // one candidate into the private table; returns false when the table is full
                    auto push = [&](const u64 key, const int mat, const int score, const int order) -> bool {
                        u32 h = hash64(key) & (u32)(LN_TC - 1);
                        const u32 klo = (u32)key, khi = (u32)(key >> 32);
                        for(int probe = 0; probe < LN_TC; probe++) {
                            const u32 lo = S.tkLo[h][lane], hi = S.tkHi[h][lane];
                            const bool free = (lo & hi) == 0xFFFFFFFFu;
                            if(free) { if(nT >= LN_TMAX) return false; S.tkLo[h][lane] = klo; S.tkHi[h][lane] = khi; nT++; }
                            if(free || (lo == klo && hi == khi)) { u32 v; pack_best<DpLane>(v, score, order); if(v > S.tb[mat][h][lane]) S.tb[mat][h][lane] = v; return true; }
                            h = (h + 1) & (u32)(LN_TC - 1);
                        }
                        return false;
                    };
                    // ---- generate: entry i of the m-2 frontier (match / mismatch, :565-607) and entry i of the m-1 frontier (gaps, :613-754).
                    // Four entries at a time: their node records (and read characters) are requested together, then the candidates are pushed -- one
                    // round trip to memory per four entries instead of one per entry (the lane's own chain of dependent loads is what bounds this class).
                    const int nMax = n1 > n2 ? n1 : n2;
                    constexpr int GU = 4;
                    for(int i0 = 0; i0 < nMax && !leave; i0 += GU) {
                        u64 pkA[GU], pkB[GU]; bool doA[GU], hasB[GU]; int pDA[GU]; u32 fa[GU], fb[GU];
                        int4 ra0[GU], rb0[GU]; int ra1w[GU], rb1w[GU]; unsigned char rc[GU];
#pragma unroll
                        for(int u = 0; u < GU; u++) {
                            const int i = i0 + u;
                            const bool hasA = i < n2; hasB[u] = i < n1;
                            pkA[u] = hasA ? (((u64)S.fkHi[1][i & (LN_FC - 1)][lane] << 32) | S.fkLo[1][i & (LN_FC - 1)][lane]) : 0;
                            pkB[u] = hasB[u] ? (((u64)S.fkHi[0][i & (LN_FC - 1)][lane] << 32) | S.fkLo[0][i & (LN_FC - 1)][lane]) : 0;
                            const int nxA = key_x(pkA[u]) + dir, nyA = key_y(pkA[u]) + dir;
                            doA[u] = hasA && !(nxA > max_levelI || nyA > seqLen || nxA < 0 || nyA < 0);
                            pDA[u] = hasA ? (int)(short)(S.fA[1][i & (LN_FC - 1)][lane] >> 16) : 0;
                            fa[u] = hasB[u] ? S.fA[0][i & (LN_FC - 1)][lane] : 0; fb[u] = hasB[u] ? S.fB[0][i & (LN_FC - 1)][lane] : 0;
                            ra0[u] = make_int4(0, 0, 0, 0); rb0[u] = ra0[u]; ra1w[u] = 0; rb1w[u] = 0; rc[u] = 0;
                            if(doA[u]) { const int nodeA = key_node(pkA[u]), pyA = key_y(pkA[u]); ra0[u] = nrec[2 * (size_t)nodeA]; ra1w[u] = ((const int*)nrec)[8 * (size_t)nodeA + 7]; rc[u] = fwd ? seqp[pyA] : seqp[pyA - 1]; }
                            if(hasB[u]) { const int nodeB = key_node(pkB[u]); rb0[u] = nrec[2 * (size_t)nodeB]; rb1w[u] = ((const int*)nrec)[8 * (size_t)nodeB + 7]; }
                        }
#pragma unroll
                        for(int u = 0; u < GU; u++) {
                            const int i = i0 + u;
                            if(leave || !(doA[u] || hasB[u])) continue;
                            const int pxA = key_x(pkA[u]), pyA = key_y(pkA[u]);
                            const int pxB = key_x(pkB[u]), pyB = key_y(pkB[u]), nodeB = key_node(pkB[u]);
                            const int nxA = pxA + dir, nyA = pyA + dir;
                            const int pD = (int)(short)(fa[u] >> 16), pG = (int)(short)(fb[u] & 0xFFFFu), pS = (int)(short)(fb[u] >> 16);
                            const int degA = ra0[u].y & 0xFFFF, degB = rb0[u].y & 0xFFFF, njB = (int)((u32)rb0[u].y >> 16);
                            // what this class does not handle: more than two edges, a gap-path jump (cells reached early)
                            if((doA[u] && degA > 2) || (hasB[u] && (degB > 2 || njB > 0))) { leave = true; continue; }
                            const int ord0 = (1 << (DpLane::IBITS + 8)) | (i << 8);
                            if(doA[u]) {
                                const unsigned char labA0 = (unsigned char)(ra1w[u] & 0xFF), labA1 = (unsigned char)((ra1w[u] >> 8) & 0xFF); const int rkA1 = (ra1w[u] >> 16) & 1;
                                if(degA > 0) if(!push(mk_key(nxA, nyA, ra0[u].z), M_D, pDA[u] + (labA0 == rc[u] ? 2 : -5), (i << 8) | 0)) leave = true;
                                if(degA > 1) if(!push(mk_key(nxA, nyA, ra0[u].w), M_D, pDA[u] + (labA1 == rc[u] ? 2 : -5), (i << 8) | rkA1)) leave = true;
                                edges += degA;
                            }
                            if(hasB[u]) {
                                const int nyG = pyB + dir, nxB = pxB + dir;
                                if(nyG >= 0 && nyG <= seqLen) {                                                     // gap in graph, :621-661
                                    if(!push(mk_key(pxB, nyG, nodeB), M_GG, pD - 6, ord0 | 0)) leave = true;
                                    if(pG != DP_NEG) if(!push(mk_key(pxB, nyG, nodeB), M_GG, pG - 2, ord0 | 1)) leave = true;
                                }
                                if(nxB >= 0 && nxB <= max_levelI) {                                                 // gap in sequence, :664-754
                                    const unsigned char labB0 = (unsigned char)(rb1w[u] & 0xFF), labB1 = (unsigned char)((rb1w[u] >> 8) & 0xFF); const int rkB1 = (rb1w[u] >> 16) & 1;
                                    for(int kk = 0; kk < 2; kk++) {
                                        if(kk >= degB) break;
                                        const unsigned char lab = kk ? labB1 : labB0; const int rk = kk ? rkB1 : 0; const u64 k = mk_key(nxB, pyB, kk ? rb0[u].w : rb0[u].z);
                                        if(lab != '_') {
                                            if(!push(k, M_SG, pD - 6, ord0 | (2 * rk))) leave = true;
                                            if(pS != DP_NEG) if(!push(k, M_SG, pS - 2, ord0 | (2 * rk + 1))) leave = true;
                                        } else {                                                                   // across a '_' edge: SG stays SG, D stays D (non-affine, :738-752)
                                            if(pS != DP_NEG) if(!push(k, M_SG, pS, ord0 | (2 * rk + 1))) leave = true;
                                            if(!push(k, M_D, pD, ord0 | rk)) leave = true;
                                        }
                                    }
                                    edges += degB;
                                }
                            }
                        }
                    }
                    if(leave) phase = LPH_LEAVE;
                    else {
                        // ---- evaluate (:840-1062): every target is a new cell (see the header); then filter (:1076-1102)
                        int itMax = DP_NEG; u64 itMaxKey = ~0ull; int itMaxSlot = -1; bool anyEqDiff = false;
                        const int nCells0 = nCells, curMax0 = curMax;
                        for(int t = 0; t < LN_TC && !leave; t++) {
                            const u32 lo = S.tkLo[t][lane], hi = S.tkHi[t][lane];
                            if((lo & hi) == 0xFFFFFFFFu) continue;
                            const u64 key = ((u64)hi << 32) | lo;
                            const u32 bD = S.tb[M_D][t][lane], bG = S.tb[M_GG][t][lane], bS = S.tb[M_SG][t][lane];
                            const int Dc = best_score<DpLane>(bD), GGv = best_score<DpLane>(bG), SGv = best_score<DpLane>(bS);
                            int Dv = Dc, dsel = 0;                       // D candidates first, then GG, then SG (:840-865); first maximum wins
                            if(GGv > Dv) { Dv = GGv; dsel = 1; }
                            if(SGv > Dv) { Dv = SGv; dsel = 2; }
                            S.tb[0][t][lane] = 0xFFFFFFFFu;              // (dropped unless kept below)
                            if(Dv < -16) continue;                       // :949
                            if(nCells >= LN_CELLS) { leave = true; break; }
                            const int slot = nCells++;
                            u32 btD = 0, btG = 0, btS = 0; int srcScore = 0;
                            if(bG) { const int o = best_order<DpLane>(bG); const int i = (o >> 8) & (LN_FC - 1); const int j = o & 255; btG = mk_bt((int)(S.fA[0][i][lane] & 0xFFFFu), j ? 1 : 0, K_GGAP, -1); }
                            if(bS) { const int o = best_order<DpLane>(bS); const int i = (o >> 8) & (LN_FC - 1); const int j = o & 255; btS = mk_bt((int)(S.fA[0][i][lane] & 0xFFFFu), (j & 1) ? 2 : 0, K_SGAP, j >> 1); }
                            if(dsel == 0) {
                                const int o = best_order<DpLane>(bD); const int ph = o >> (DpLane::IBITS + 8); const int i = (o >> 8) & (LN_FC - 1); const int j = o & 255;
                                const u32 fa = S.fA[ph ? 0 : 1][i][lane];
                                srcScore = (int)(short)(fa >> 16);
                                btD = mk_bt((int)(fa & 0xFFFFu), 0, ph ? K_SGAP : K_DIAG, j);
                            } else if(dsel == 1) {
                                btD = mk_bt(slot, 1, K_HOP, -1);
                                const int o = best_order<DpLane>(bG); const int i = (o >> 8) & (LN_FC - 1); const int j = o & 255;
                                srcScore = j ? (int)(short)(S.fB[0][i][lane] & 0xFFFFu) : (int)(short)(S.fA[0][i][lane] >> 16);
                            } else {
                                btD = mk_bt(slot, 2, K_HOP, -1);
                                const int o = best_order<DpLane>(bS); const int i = (o >> 8) & (LN_FC - 1); const int j = o & 255;
                                srcScore = (j & 1) ? (int)(short)(S.fB[0][i][lane] >> 16) : (int)(short)(S.fA[0][i][lane] >> 16);
                            }
                            uint4* dst = (uint4*)(cells + slot);
                            dst[0] = make_uint4(lo, hi, ((u32)(unsigned short)(short)Dv) | ((u32)(unsigned short)(short)GGv << 16), (u32)(unsigned short)(short)SGv);
                            dst[1] = make_uint4(btD, btG, btS, 0u);
                            if(key_y(key) == limitY) { if(nCompleted >= LN_COMPLETED) { leave = true; break; } completed[nCompleted++] = slot | ((Dv + 64) << 16); }      // :982-999 (with its D score: a cell of this class is never overwritten)
                            // running maximum bookkeeping, :1007-1062 (the real previous step of D is a frontier cell: its cached score)
                            if(Dv == curMax0 && (Dv - srcScore) != 0) anyEqDiff = true;
                            if(Dv > itMax || (Dv == itMax && key < itMaxKey)) { itMax = Dv; itMaxKey = key; itMaxSlot = slot; }
                            // kept: stash slot | D and GG | SG for the filter
                            S.tb[0][t][lane] = (u32)slot | ((u32)(unsigned short)(short)Dv << 16);
                            S.tb[1][t][lane] = ((u32)(unsigned short)(short)GGv) | ((u32)(unsigned short)(short)SGv << 16);
                        }
                        if(leave) phase = LPH_LEAVE;
                        else {
                            cellsEvaluated += (u32)nT;
                            if(itMax > curMax0) { curMax = itMax; lastInc = d; firstMaxSlot = itMaxSlot; }
                            if(anyEqDiff) lastInc = d;
                            // the last diagonal becomes the one before; survivors of the X-drop window (15 below the iteration's maximum) form the new one, in key order
#pragma unroll
                            for(int q = 0; q < LN_FC; q++) { S.fkLo[1][q][lane] = S.fkLo[0][q][lane]; S.fkHi[1][q][lane] = S.fkHi[0][q][lane]; S.fA[1][q][lane] = S.fA[0][q][lane]; S.fB[1][q][lane] = S.fB[0][q][lane]; }
                            int nNew = 0;
                            for(int t = 0; t < LN_TC; t++) {
                                const u32 lo = S.tkLo[t][lane], hi = S.tkHi[t][lane];
                                if((lo & hi) == 0xFFFFFFFFu) continue;
                                const u32 a = S.tb[0][t][lane], bq = S.tb[1][t][lane];
                                S.tkLo[t][lane] = 0xFFFFFFFFu; S.tkHi[t][lane] = 0xFFFFFFFFu; S.tb[0][t][lane] = 0; S.tb[1][t][lane] = 0; S.tb[2][t][lane] = 0;
                                if(a == 0xFFFFFFFFu) continue;
                                if(itMax - (int)(short)(a >> 16) > 15) continue;
                                if(nNew >= LN_FC) { leave = true; continue; }
                                // insertion by key (ascending): the new diagonal is pushed in std::map order next iteration
                                const u64 key = ((u64)hi << 32) | lo;
                                int pos = nNew;
                                while(pos > 0) {
                                    const u64 pk = ((u64)S.fkHi[0][pos - 1][lane] << 32) | S.fkLo[0][pos - 1][lane];
                                    if(pk < key) break;
                                    S.fkLo[0][pos][lane] = S.fkLo[0][pos - 1][lane]; S.fkHi[0][pos][lane] = S.fkHi[0][pos - 1][lane]; S.fA[0][pos][lane] = S.fA[0][pos - 1][lane]; S.fB[0][pos][lane] = S.fB[0][pos - 1][lane];
                                    pos--;
                                }
                                S.fkLo[0][pos][lane] = lo; S.fkHi[0][pos][lane] = hi; S.fA[0][pos][lane] = a; S.fB[0][pos][lane] = bq;
                                nNew++;
                            }
                            if(leave) phase = LPH_LEAVE;
                            else { n2 = n1; n1 = nNew; itersRun = d; d++; }
                        }
                    }
                }
            }

            // ================= a call that this class cannot finish: on to the 16-lane class, no trace left here =================
            bool leaving = phase == LPH_LEAVE;
            if(phase == LPH_LEAVE) {
                // (the table may hold candidates of the abandoned iteration)
#pragma unroll
                for(int t = 0; t < LN_TC; t++) { S.tkLo[t][lane] = 0xFFFFFFFFu; S.tkHi[t][lane] = 0xFFFFFFFFu; S.tb[0][t][lane] = 0; S.tb[1][t][lane] = 0; S.tb[2][t][lane] = 0; }
                phase = LPH_IDLE;
            }

            // ================= end cell, backtrace, columns (:1381-1517, :1109-1354), for the call and for the calls that share it =================
            // Finishing is long and serial per lane: lanes wait until a fair number of them are ready (or nothing else is left to run), then finish together.
            {
                const u64 fin = __ballot(phase == LPH_FINISH), run = __ballot(phase == LPH_RUN);
                if(fin && (__popcll(fin) >= 24 || run == 0)) {
                    if(phase == LPH_FINISH) {
                        const uint8_t* seqp = readBases + rOff;
                        const int stride = B.stride;
                        int cur = item; bool leave = false;
                        // results of all copies are produced first and written only if none of them has to leave (a copy that outgrows the class takes the DP along)
                        while(cur >= 0 && !leave) {
                            int endSlot = -1, endScore = 0;
                            if(nCompleted > 0) {
                                int best = DP_NEG;
                                for(int i = 0; i < nCompleted; i++) best = max(best, (completed[i] >> 16) - 64);
                                int nTies = 0;
                                for(int i = 0; i < nCompleted; i++) if((completed[i] >> 16) - 64 == best) nTies++;
                                u32 sd = rng_seed + (u32)cur;
                                const int selectedIndex = glibc_rand_r(&sd) % nTies;                              // Utilities.cpp:922-927
                                int found = -1;
                                for(int i = 0; i < nCompleted && found < 0; i++) {
                                    const int s = completed[i] & 0xFFFF;
                                    if((completed[i] >> 16) - 64 != best) continue;
                                    int rank = 0;
                                    if(nTies > 1) {
                                        const u64 k = cells[s].key; const int kz = key_node(k) - G.level_off[key_x(k)];
                                        for(int u = 0; u < nCompleted; u++) { const int su = completed[u] & 0xFFFF; if(su != s && (completed[u] >> 16) - 64 == best) { const u64 ku = cells[su].key;
                                            if(xz_less(key_x(ku), key_node(ku) - G.level_off[key_x(ku)], key_x(k), kz)) rank++; } }
                                    }
                                    if(rank == selectedIndex) found = s;
                                }
                                endSlot = found; endScore = best;
                            } else if(curMax > 0) { endSlot = firstMaxSlot; endScore = curMax; }      // (the first cell of the running maximum holds it: nothing is overwritten here)
                            int have = 0, sb = 0, se = -1, nCols = 0, err = 0;
                            if(endSlot >= 0) {
                                const u64 ek = cells[endSlot].key; const int yEnd = key_y(ek);
                                if(fwd) { sb = start_seq; se = yEnd - 1; } else { sb = yEnd; se = start_seq - 1; }      // toVerboseSeedChain, VirtualNWUnique.cpp:28-29
                                // Backtrace and columns in one pass: the column of a step needs the node of the PREVIOUS cell (its edge is the rank-th one from that node to
                                // this cell's node), which the chase loads next anyway -- so step k's column is written while step k + 1's cell is on its way.  The left
                                // extension is written at its final place [sb, sb + n); the right extension right-aligned in the row, i.e. step s at stride - 1 - s, which
                                // is where the reversal of a forward trace (:1319-1326) puts it whatever its length (k_stitch_chains moves it next to the seed).
                                const int c = cur >> 1; const size_t rowBase = row_base(B, c);
                                int* oL = B.ext_level + rowBase; int* oE = B.ext_edge + rowBase; uint8_t* oG = B.ext_g + rowBase; uint8_t* oS = B.ext_s + rowBase;
                                const int* eo = fwd ? G.out_off : G.in_off; const int* et = fwd ? G.out_to : G.in_from; const int* ee = fwd ? G.out_eid : G.in_eid;
                                int slot = endSlot, m = 0, x = key_x(ek), y = yEnd, nSteps = 0, guard = 0;
                                bool pend = false; u32 pendB = 0; u64 pendKey = 0; int pendS = 0;        // the step whose column waits for its previous cell
                                bool bad = false;
                                auto emit = [&](const u32 b, const u64 xy, const int sIdx, const int pnode) {
                                    const int pos = fwd ? stride - 1 - sIdx : sb + sIdx;
                                    if(pos < 0 || pos >= stride) return;                                  // (a trace longer than the row: reported below / by k_stitch_chains)
                                    const int kind = bt_kind(b); const int xx = key_x(xy), yy = key_y(xy);
                                    const unsigned char sc = fwd ? (yy >= 1 ? seqp[yy - 1] : 0) : (yy < seqLen ? seqp[yy] : 0);
                                    if(kind == K_GGAP) { oL[pos] = -1; oE[pos] = -1; oG[pos] = '_'; oS[pos] = sc; return; }
                                    const int node = key_node(xy); int j = bt_edge(b);
                                    const int q1 = eo[pnode + 1]; int q = eo[pnode];
                                    for(; q < q1; q++) if(et[q] == node && j-- == 0) break;
                                    if(q >= q1) { bad = true; return; }                                   // (cannot happen: the rank was derived from these very arrays)
                                    const int eid = ee[q];
                                    oL[pos] = fwd ? xx - 1 : xx; oE[pos] = eid; oG[pos] = G.edge_label[eid]; oS[pos] = (kind == K_DIAG) ? sc : (unsigned char)'_';
                                };
                                while((x != startLevel || y != start_seq) && guard++ < 4 * LN_STEPS) {
                                    const CellRec* cr = cells + slot;
                                    const u32 b = cr->bt[m]; const u64 ckey = cr->key;
                                    const int kind = bt_kind(b);
                                    if(kind != K_HOP) {
                                        if(pend) { emit(pendB, pendKey, pendS, key_node(ckey)); pend = false; }       // this cell IS the previous cell of the pending step
                                        if(nSteps >= LN_STEPS) { leave = true; break; }
                                        pendB = b; pendKey = ckey; pendS = nSteps; pend = true; nSteps++;
                                    }
                                    if(kind == K_DIAG) { x -= dir; y -= dir; } else if(kind == K_GGAP) { y -= dir; } else if(kind == K_SGAP) { x -= dir; }
                                    slot = bt_prev(b); m = bt_src(b);
                                }
                                if(guard >= 4 * LN_STEPS) leave = true;
                                if(pend && !leave) emit(pendB, pendKey, pendS, key_node(cells[slot].key));            // the last step: its previous cell is the start cell
                                if(bad) leave = true;
                                nCols = nSteps;
                                if(!leave) {
                                    if(nCols > stride) err = -1000000 - nCols;
                                    else if(sb > se) leave = true;               // (the group classes raise this as a capacity failure: let them)
                                    else have = 1;
                                }
                            }
                            if(leave) break;
                            B.dp_iters[cur] = itersRun; B.dp_score[cur] = have ? endScore : INT32_MIN;
                            B.dp_ncols[cur] = have ? nCols : -1; B.dp_sb[cur] = sb; B.dp_se[cur] = se; B.dp_err[cur] = err;
                            accCalls++; accIters += (u64)itersRun; accCells += (u64)cellsEvaluated; accEdges += (u64)edges;
                            // the chains of the read whose DP starts from the same cell (k_dp_items): same iterations, their own end-cell draw
                            int nx = B.dp_alias_head[cur]; if(nx < 0) nx = B.dp_alias_next[cur];     // the DP that ran heads the list, a duplicate is on it
                            cur = nx;
                        }
                        if(leave) {
                            // a backtrace longer than this class holds: the 16-lane class runs the call again.  Copies finished above keep their (identical) results;
                            // the item entry is handed to the copy that is still to do (as the group classes do for a linked duplicate)
                            if(cur != item) ((int*)(items + itemIdx))[0] = cur;
                            leaving = true;
                        }
                        phase = LPH_IDLE;
                    }
                }
            }
            // ---- the items that left in this trip: into the wave's LDS list, queued for the 16-lane class 64 at a time
            {
                const u64 lv = __ballot(leaving);
                if(lv) {
                    if(leaving) outBuf[nOut + (int)__popcll(lv & ((1ull << lane) - 1ull))] = itemIdx;
                    nOut += (int)__popcll(lv);
                    WSYNC();
                    if(nOut >= 64) {
                        int base = 0; if(lane == 0) base = atomicAdd(tinyCount, 64); base = __builtin_amdgcn_readfirstlane(base);
                        tinyOut[base + lane] = outBuf[lane];
                        const int rest = nOut - 64;
                        const int mv = lane < rest ? outBuf[64 + lane] : 0;
                        WSYNC();
                        if(lane < rest) outBuf[lane] = mv;
                        nOut = rest;
                        WSYNC();
                    }
                }
            }
        }
        if(nOut > 0) {        // what is left of the wave's list
            int base = 0; if(lane == 0) base = atomicAdd(tinyCount, nOut); base = __builtin_amdgcn_readfirstlane(base);
            if(lane < nOut) tinyOut[base + lane] = outBuf[lane];
            WSYNC();
        }
    }
    // work counters: one set of atomics per wave (a lane's share of a launch is a few thousand iterations: the wave's sums fit 31 bits)
    const int wCalls = wave_sum_i32((int)accCalls), wIters = wave_sum_i32((int)accIters), wCells = wave_sum_i32((int)accCells), wEdges = wave_sum_i32((int)accEdges);
    if(lane == 0 && wCalls) { atomicAdd(&B.counters[CNT_DP_CALLS], (u64)wCalls); atomicAdd(&B.counters[CNT_DP_ITERS], (u64)wIters); atomicAdd(&B.counters[CNT_DP_CELLS], (u64)wCells); atomicAdd(&B.counters[CNT_EDGES], (u64)wEdges); }
}

}  // namespace hlala
