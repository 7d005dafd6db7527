// kernel_extend.hip -- stage B: extensionAligner::extendSeedChain + scoreOneAlignment on gfx950.
//
// One 64-lane wavefront per chain (block = 64 threads).  The affine X-drop frontier DP of
// fullNeedleman_diagonal_extension_gapJumper (mapper/aligner/extensionAligner.cpp:335-1556) runs
// as a sequence of wave-synchronous phases per iteration:
//   generate : one lane per frontier cell pushes its candidates into an LDS hash keyed by the
//              target cell; ties are resolved with ds_max_u32 on (score, reversed push index), which
//              is exactly the reference's "first maximum in push order" (Utilities.cpp:379-406)
//   evaluate : one lane per target cell combines the three matrices, applies the -16 keep threshold,
//              merges into the cell table (HBM scratch slab private to the wave) and derives the
//              running-maximum / patience bookkeeping with wave reductions
//   filter   : X-drop window of 15 below the iteration maximum, then a rank sort by (x,y,z) so the next
//              iteration pushes in the reference's std::map order.
// Scores are integers (the reference's doubles only ever hold integers, alignerBase.cpp:19-25).
#include "batch.h"
#include "../../include/hlala_gpu.h"

namespace hlala {

enum { K_DIAG = 0, K_GGAP = 1, K_SGAP = 2, K_HOP = 3, K_JUMP = 4 };
enum { M_D = 0, M_GG = 1, M_SG = 2 };

constexpr u64 HKEY_EMPTY = ~0ull;
constexpr int DP_IMPCAP = 2048;    // staged improvements of existing cells per iteration (rare path, kept in the slab)

// Two capacity classes: the small one serves almost every chain at 6-7 blocks per CU; chains whose frontier /
// candidate set outgrows it (long runs of parallel gap paths) are re-run by the large one (one block per CU).
struct DpSmall { static constexpr int WCAP = 64,      HC = 128,   IBITS = 7,  SEQCAP = 512;       typedef u32 Best; };
struct DpLarge { static constexpr int WCAP = 1024,    HC = 2048,  IBITS = 10, SEQCAP = DP_SEQCAP; typedef u64 Best; };

template <class C>
struct __align__(16) DpLdsT {
    u64 hkey[C::HC];
    typename C::Best hbest[3][C::HC];
    unsigned short tlist[C::HC];
    u64 fkey[3][C::WCAP];
    short fslot[3][C::WCAP];        // table slot of the frontier cell
    short fD[3][C::WCAP], fG[3][C::WCAP], fS[3][C::WCAP];
    unsigned char seq[C::SEQCAP];
    short tes[C::HC];               // per target: existing / assigned table slot (-1 = none; DP_CELLS <= 32767)
    unsigned char timp[C::HC];      // per target: improved-matrix mask | 0x80 = new cell
    int nT, nNew, nImp, nKeepF, err, nCompletedAdd;
};

struct ExtSlab {
    u64* cell_key;      // [DP_CELLS]
    short* cell_sc;     // [DP_CELLS*4]  D, GG, SG, -
    u64* cell_bt;       // [3*DP_CELLS]
    u64* early_key;     // [DP_EARLY]
    int* early_val;     // [DP_EARLY]
    int* completed;     // [DP_COMPLETED]
    u64* step_bt;       // [DP_STEPS]
    u64* step_xy;       // [DP_STEPS]
    // existing table entries strictly improved in the current iteration (staged, :951-979)
    int* imp_slot; u64* imp_key; short* imp_new; u64* imp_bt; int* imp_mask;      // [DP_IMPCAP] (x4 / x3 for new / bt)
    int* x_level[2];    // [stride] left / right extension columns
    int* x_edge[2];
    uint8_t* x_g[2];
    uint8_t* x_s[2];
};

__host__ __device__ inline size_t ext_slab_bytes(int stride)
{
    size_t b = 0;
    b += (size_t)DP_CELLS * 8 + (size_t)DP_CELLS * 8 + (size_t)DP_CELLS * 24;
    b += (size_t)DP_EARLY * 8 + (size_t)DP_EARLY * 4 + (size_t)DP_COMPLETED * 4;
    b += (size_t)DP_STEPS * 16;
    b += (size_t)DP_IMPCAP * (4 + 8 + 8 + 24 + 4);
    b += 2 * ((size_t)stride * 8 + (size_t)((stride + 7) / 8) * 8 * 2);
    return (b + 255) & ~(size_t)255;
}

__device__ inline ExtSlab ext_slab_at(char* base, int stride)
{
    ExtSlab s;
    char* p = base;
    s.cell_key = (u64*)p; p += (size_t)DP_CELLS * 8;
    s.cell_bt = (u64*)p; p += (size_t)DP_CELLS * 24;
    s.early_key = (u64*)p; p += (size_t)DP_EARLY * 8;
    s.step_bt = (u64*)p; p += (size_t)DP_STEPS * 8;
    s.step_xy = (u64*)p; p += (size_t)DP_STEPS * 8;
    s.imp_key = (u64*)p; p += (size_t)DP_IMPCAP * 8;
    s.imp_bt = (u64*)p; p += (size_t)DP_IMPCAP * 24;
    s.imp_new = (short*)p; p += (size_t)DP_IMPCAP * 8;
    s.cell_sc = (short*)p; p += (size_t)DP_CELLS * 8;
    s.early_val = (int*)p; p += (size_t)DP_EARLY * 4;
    s.completed = (int*)p; p += (size_t)DP_COMPLETED * 4;
    s.imp_slot = (int*)p; p += (size_t)DP_IMPCAP * 4;
    s.imp_mask = (int*)p; p += (size_t)DP_IMPCAP * 4;
    size_t s8 = (size_t)((stride + 7) / 8) * 8;
    for(int k = 0; k < 2; k++) {
        s.x_level[k] = (int*)p; p += (size_t)stride * 4;
        s.x_edge[k] = (int*)p; p += (size_t)stride * 4;
        s.x_g[k] = (uint8_t*)p; p += s8;
        s.x_s[k] = (uint8_t*)p; p += s8;
    }
    return s;
}

// DP cell key: level x (24 bits) | read offset y (12 bits) | node id (28 bits).  Node ids are level-major and
// stable in creation order, so unsigned key order == the reference's std::map order (x, then y, then rank z).
__device__ __forceinline__ u64 mk_key(int x, int y, int node) { return ((u64)(u32)x << 40) | ((u64)(u32)y << 28) | (u64)(u32)node; }
__device__ __forceinline__ int key_x(u64 k) { return (int)(k >> 40); }
__device__ __forceinline__ int key_y(u64 k) { return (int)((k >> 28) & 0xFFF); }
__device__ __forceinline__ int key_node(u64 k) { return (int)(k & 0xFFFFFFF); }
// targets of one iteration differ in a few low bits of node / y / x: one 32-bit multiply spreads them
__device__ __forceinline__ u32 hash64(u64 k) { u32 h = (u32)k ^ ((u32)(k >> 28) * 0x9E3779B1u) ^ ((u32)(k >> 40) * 0x85EBCA6Bu); h *= 0x9E3779B1u; return h ^ (h >> 15); }

__device__ __forceinline__ u64 mk_bt(int prev, int src, int kind, int edge) { return ((u64)(u32)edge << 32) | (u64)((u32)prev | ((u32)src << 24) | ((u32)kind << 26)); }
__device__ __forceinline__ int bt_prev(u64 b) { return (int)(b & 0xFFFFFF); }
__device__ __forceinline__ int bt_src(u64 b) { return (int)((b >> 24) & 3); }
__device__ __forceinline__ int bt_kind(u64 b) { return (int)((b >> 26) & 7); }
__device__ __forceinline__ int bt_edge(u64 b) { return (int)(b >> 32); }

// candidate value: (score, reversed push index) so that an unsigned max = highest score, earliest push
__device__ __forceinline__ void pack_best(u32& o, int score, int order) { o = ((u32)(score + 64) << 16) | (u32)(0xFFFF - order); }
__device__ __forceinline__ void pack_best(u64& o, int score, int order) { o = ((u64)(u32)(score + 64) << 32) | (u64)(u32)(0x7FFFFFFF - order); }
__device__ __forceinline__ int best_score(u32 b) { return b ? (int)(b >> 16) - 64 : DP_NEG; }
__device__ __forceinline__ int best_score(u64 b) { return b ? (int)(b >> 32) - 64 : DP_NEG; }
__device__ __forceinline__ int best_order(u32 b) { return 0xFFFF - (int)(b & 0xFFFF); }
__device__ __forceinline__ int best_order(u64 b) { return 0x7FFFFFFF - (int)(b & 0xFFFFFFFFull); }

// push one candidate (Alt::{D,GG,SG}.push_back in the reference) -- returns false on hash overflow
template <class C>
__device__ inline bool dp_push(DpLdsT<C>& S, u64 key, int mat, int score, int order)
{
    u32 h = hash64(key) & (C::HC - 1);
    for(int probe = 0; probe < C::HC; probe++) {
        u64 cur = S.hkey[h];
        if(cur == HKEY_EMPTY) {
            u64 old = atomicCAS(&S.hkey[h], HKEY_EMPTY, key);
            if(old == HKEY_EMPTY) {
                int pos = atomicAdd(&S.nT, 1);
                if(pos < C::HC) S.tlist[pos] = (unsigned short)h;
                cur = key;
            } else cur = old;
        }
        if(cur == key) { typename C::Best v; pack_best(v, score, order); atomicMax(&S.hbest[mat][h], v); return true; }
        h = (h + 1) & (C::HC - 1);
    }
    return false;
}

__device__ inline int early_lookup(const ExtSlab& sl, u64 key)
{
    u32 h = hash64(key) & (DP_EARLY - 1);
    for(int probe = 0; probe < DP_EARLY; probe++) {
        // entries are published with L2 atomics: read them past the CU's L1
        u64 cur = __hip_atomic_load(&sl.early_key[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if(cur == key) return __hip_atomic_load(&sl.early_val[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if(cur == HKEY_EMPTY) return -1;
        h = (h + 1) & (DP_EARLY - 1);
    }
    return -1;
}
__device__ inline bool early_insert(const ExtSlab& sl, u64 key, int slot)
{
    u32 h = hash64(key) & (DP_EARLY - 1);
    for(int probe = 0; probe < DP_EARLY; probe++) {
        u64 old = atomicCAS(&sl.early_key[h], HKEY_EMPTY, key);
        if(old == HKEY_EMPTY || old == key) { __hip_atomic_store(&sl.early_val[h], slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return true; }
        h = (h + 1) & (DP_EARLY - 1);
    }
    return false;
}

// "x/z" string order of std::set<std::string> achieved_complete_sequence_alignments (extensionAligner.cpp:493, 1431)
__device__ inline int render_xz(int x, int z, char* buf)
{
    char tmp[12]; int n = 0, len = 0;
    if(x == 0) tmp[n++] = '0';
    while(x > 0) { tmp[n++] = (char)('0' + x % 10); x /= 10; }
    while(n > 0) buf[len++] = tmp[--n];
    buf[len++] = '/';
    if(z == 0) tmp[n++] = '0';
    while(z > 0) { tmp[n++] = (char)('0' + z % 10); z /= 10; }
    while(n > 0) buf[len++] = tmp[--n];
    return len;
}
__device__ inline bool xz_less(int x1, int z1, int x2, int z2)
{
    char a[28], b[28];
    int la = render_xz(x1, z1, a), lb = render_xz(x2, z2, b);
    int n = la < lb ? la : lb;
    for(int i = 0; i < n; i++) { if((unsigned char)a[i] != (unsigned char)b[i]) return (unsigned char)a[i] < (unsigned char)b[i]; }
    return la < lb;
}

struct DpResult { int have, ncols, seq_begin, seq_end, iters, score, err; u64 cells; int edges; };

#define DBGW(i, v) __hip_atomic_store(&dbg[i], (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)
#define DBGB(i, v) do { if(dbg && lane == 0 && blockIdx.x < 2000) __hip_atomic_store(&dbg[64 + 4 * blockIdx.x + (i)], (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); } while(0)
#define DP_FAIL(code) do { if(lane == 0 && S.err == 0) S.err = (code); } while(0)

// extensionAligner::fullNeedleman_diagonal_extension_gapJumper (extensionAligner.cpp:335-1556) with
// returnGlobalScore = false, preferSequenceCompleAlignments = true, empty blockedPathsTable,
// diagonal_stop_threshold = -16 (the only configuration extendSeedChain uses, :229-241, :281-293).
template <class C>
__device__ DpResult dp_run(const DevGraph& G, DpLdsT<C>& S, const ExtSlab& sl, int seqLen, int start_seq, int startLevel,
                           int startNode, bool fwd, u32 seed, int side, int outCap, u64* counters, int* dbg, int chain)
{
    const int lane = lane_id();
    const int dir = fwd ? 1 : -1;
    const int max_levelI = G.L - 1, max_seqI = seqLen;       // :431-463 (min_* are 0 in both directions)
    const int limitY = fwd ? seqLen : 0;
    const long long diagonals = (long long)seqLen + G.L - 1;
    DpResult R; R.have = 0; R.ncols = 0; R.seq_begin = 0; R.seq_end = -1; R.iters = 0; R.score = INT32_MIN; R.err = 0; R.cells = 0; R.edges = 0;

    // ---- init, :480-519
    for(int i = lane; i < C::HC; i += 64) { S.hkey[i] = HKEY_EMPTY; S.hbest[0][i] = 0; S.hbest[1][i] = 0; S.hbest[2][i] = 0; }
    if(lane == 0) {
        S.nT = 0; S.err = 0;
        sl.cell_key[0] = mk_key(startLevel, start_seq, startNode);
        sl.cell_sc[0] = 0; sl.cell_sc[1] = (short)DP_NEG; sl.cell_sc[2] = (short)DP_NEG; sl.cell_sc[3] = 0;
        sl.cell_bt[0] = 0; sl.cell_bt[DP_CELLS] = 0; sl.cell_bt[2 * DP_CELLS] = 0;
        S.fkey[0][0] = mk_key(startLevel, start_seq, startNode); S.fslot[0][0] = 0;
        S.fD[0][0] = 0; S.fG[0][0] = (short)DP_NEG; S.fS[0][0] = (short)DP_NEG;
    }
    WSYNC();
    int b1 = 0, b2 = 1, bn = 2;
    int n1 = 1, n2 = 0;
    int nCells = 1, nCompleted = 0;
    int curMax = 0, firstMaxSlot = 0, lastInc = 0;
    bool earlyInit = false;
    long long itersRun = 0;
    u64 cellsEvaluated = 0, edgesTouched = 0;
    long long tGen = 0, tEval = 0, tFilt = 0, tMark = 0;
    const bool timing = dbg != nullptr;
    if(timing) tMark = clock64();

    for(long long d = 1; d <= diagonals; d++) {                                            // :531
        if((d - lastInc) > 40) break;                                                      // :553 maximum_steps_nonIncrease
        if(n1 == 0 && n2 == 0) {
            // both frontiers empty: the remaining iterations of the reference loop are no-ops
            long long last = (long long)lastInc + 40; if(last > diagonals) last = diagonals;
            itersRun = last; break;
        }
        itersRun = d;
        if(d > 60000) { DP_FAIL(__LINE__); break; }          // watchdog: far beyond any read length + patience

        // ================= generate =====================================================
        // from the m-2 diagonal: match / mismatch, :565-607
        for(int i = lane; i < n2; i += 64) {
            u64 pk = S.fkey[b2][i]; int px = key_x(pk), py = key_y(pk), node = key_node(pk);
            int nx = px + dir, ny = py + dir;
            if(nx > max_levelI || ny > max_seqI || nx < 0 || ny < 0) continue;
            unsigned char rc = fwd ? S.seq[py] : S.seq[py - 1];
            int e0 = fwd ? G.out_off[node] : G.in_off[node], e1 = fwd ? G.out_off[node + 1] : G.in_off[node + 1];
            int pD = S.fD[b2][i];
            if(e1 - e0 > 127) { S.err = __LINE__; continue; }
            for(int e = e0; e < e1; e++) {
                int tn = fwd ? G.out_to[e] : G.in_from[e];
                unsigned char lab = fwd ? G.out_label[e] : G.in_label[e];
                int sc = pD + (lab == rc ? 2 : -5);
                if(!dp_push<C>(S, mk_key(nx, ny, tn), M_D, sc, (i << 8) | (e - e0))) S.err = __LINE__;
            }
            edgesTouched += (u64)(e1 - e0);
        }
        // from the m-1 diagonal: gaps and jumps, :613-787
        for(int i = lane; i < n1; i += 64) {
            u64 pk = S.fkey[b1][i]; int px = key_x(pk), py = key_y(pk), node = key_node(pk);
            int pD = S.fD[b1][i], pG = S.fG[b1][i], pS = S.fS[b1][i];
            int ord0 = (1 << (C::IBITS + 8)) | (i << 8);
            {   // gap in graph, :621-661
                int ny = py + dir;
                if(ny >= 0 && ny <= max_seqI) {
                    u64 k = mk_key(px, ny, node);
                    if(!dp_push<C>(S, k, M_GG, pD - 6, ord0 | 0)) S.err = __LINE__;
                    if(pG != DP_NEG) if(!dp_push<C>(S, k, M_GG, pG - 2, ord0 | 1)) S.err = __LINE__;
                }
            }
            int deg = 0;
            {   // gap in sequence, :664-754
                int nx = px + dir;
                int e0 = fwd ? G.out_off[node] : G.in_off[node], e1 = fwd ? G.out_off[node + 1] : G.in_off[node + 1];
                deg = e1 - e0;
                if(deg > 127) { S.err = __LINE__; continue; }
                if(nx >= 0 && nx <= max_levelI) {
                    for(int e = e0; e < e1; e++) {
                        int tn = fwd ? G.out_to[e] : G.in_from[e];
                        unsigned char lab = fwd ? G.out_label[e] : G.in_label[e];
                        u64 k = mk_key(nx, py, tn);
                        int kk = e - e0;
                        if(lab != '_') {
                            if(!dp_push<C>(S, k, M_SG, pD - 6, ord0 | (2 * kk))) S.err = __LINE__;
                            if(pS != DP_NEG) if(!dp_push<C>(S, k, M_SG, pS - 2, ord0 | (2 * kk + 1))) S.err = __LINE__;
                        } else {
                            if(pS != DP_NEG) if(!dp_push<C>(S, k, M_SG, pS, ord0 | (2 * kk + 1))) S.err = __LINE__;
                            if(!dp_push<C>(S, k, M_D, pD, ord0 | kk)) S.err = __LINE__;         // non-affine sequence gap, :738-752
                        }
                    }
                    edgesTouched += (u64)deg;
                }
            }
            {   // gap-path jumps, :757-786 (jump_length * S_graphGap = 0)
                // push index of a jump = 128 + its rank in the jump table: after every edge candidate of the same source (:757)
                const int* joff = fwd ? G.jf_off : G.jb_off; const int* jnode = fwd ? G.jf_node : G.jb_node; const int* jlvl = fwd ? G.jf_lvl : G.jb_lvl;
                int j0 = joff[node], j1 = joff[node + 1];
                if(j1 - j0 > 127) { S.err = __LINE__; continue; }
                for(int j = j0; j < j1; j++) {
                    int tn = jnode[j]; int jx = jlvl[j];
                    if(jx < 0 || jx > max_levelI) continue;
                    if(!dp_push<C>(S, mk_key(jx, py, tn), M_D, pD, ord0 | (128 + (j - j0)))) S.err = __LINE__;
                }
            }
        }
        WSYNC();
        if(timing) { long long t = clock64(); tGen += t - tMark; tMark = t; }
        int nT = uni(S.nT);
        if(nT > (C::HC * 3) / 4 || uni(S.err)) { DP_FAIL(__LINE__); break; }
        cellsEvaluated += (u64)nT;

        // ================= evaluate =====================================================
        if(lane == 0) { S.nNew = 0; S.nImp = 0; S.nCompletedAdd = 0; }
        WSYNC();
        int itMaxNew = DP_NEG;        // max Dv over kept targets of this iteration
        u64 itMaxKey = ~0ull;         // smallest key achieving it (= first such cell in std::map order)
        bool anyEqDiff = false, anyOw = false, anyExisting = false;
        const int curMax0 = curMax;

        // Cells reached through a gap-path jump arrive EARLIER than their natural diagonal |dx|+|dy| and can be
        // reached again later ("scores" merge, :951-979).  Only such cells are registered in the early hash, and
        // only while it is non-empty do targets need an existence lookup.
        if(earlyInit) {
            for(int t0 = 0; t0 < nT; t0 += 64) {
                int t = t0 + lane; int es = -1;
                if(t < nT) {
                    int h = S.tlist[t];
                    int Dv = max(best_score(S.hbest[M_D][h]), max(best_score(S.hbest[M_GG][h]), best_score(S.hbest[M_SG][h])));
                    if(Dv >= -16) es = early_lookup(sl, S.hkey[h]);
                    S.tes[t] = (short)es;
                }
                if(__ballot(es >= 0)) anyExisting = true;
            }
            WSYNC();
        }
        const bool slow = anyExisting;
        const bool hadEarly = earlyInit;      // S.tes[] holds lookups only if the pre-pass ran

        for(int pass = 0; pass < (slow ? 2 : 1); pass++) {
            for(int t0 = 0; t0 < nT; t0 += 64) {
                int t = t0 + lane;
                bool act = t < nT;
                int h = act ? S.tlist[t] : 0;
                u64 key = act ? S.hkey[h] : 0;
                typename C::Best bD = act ? S.hbest[M_D][h] : 0, bG = act ? S.hbest[M_GG][h] : 0, bS = act ? S.hbest[M_SG][h] : 0;
                int Dc = best_score(bD), GGv = best_score(bG), SGv = best_score(bS);
                int Dv = Dc, dsel = 0;                      // D candidates first, then GG, then SG (:840-865); first maximum wins
                if(GGv > Dv) { Dv = GGv; dsel = 1; }
                if(SGv > Dv) { Dv = SGv; dsel = 2; }
                bool keep = act && (Dv >= -16);                                               // :949
                int es = -1; bool isNew; int slot;
                if(pass == 0) {
                    if(hadEarly && keep) es = S.tes[t];
                    isNew = keep && es < 0;
                    int total; int off = wave_excl_scan(isNew ? 1 : 0, total);
                    slot = isNew ? nCells + off : es;
                    if(nCells + total > DP_CELLS) { DP_FAIL(__LINE__); }
                    nCells += total;
                } else {
                    slot = keep ? S.tes[t] : -1;
                    isNew = keep && (S.timp[t] & 0x80);
                    es = isNew ? -1 : slot;
                }
                // ---- back pointers of the three matrices, decoded from the winning push index
                u64 btD = 0, btG = 0, btS = 0;
                int srcScore = 0;       // score the real previous step came from (fast form of the `diff` rule)
                if(keep && S.err == 0 && slot >= 0 && slot < DP_CELLS) {
                    // back pointer = (previous cell slot, source matrix, kind, local push index j); the graph edge / gap path behind j
                    // is resolved only for the cells on the final path, at backtrace time
                    if(bG) { int o = best_order(bG); int i = (o >> 8) & ((1 << C::IBITS) - 1); int j = o & 255;
                             btG = mk_bt(S.fslot[b1][i], j ? 1 : 0, K_GGAP, -1); }
                    if(bS) { int o = best_order(bS); int i = (o >> 8) & ((1 << C::IBITS) - 1); int j = o & 255;
                             btS = mk_bt(S.fslot[b1][i], (j & 1) ? 2 : 0, K_SGAP, j >> 1); }
                    if(dsel == 0) {
                        int o = best_order(bD); int ph = o >> (C::IBITS + 8); int i = (o >> 8) & ((1 << C::IBITS) - 1); int j = o & 255;
                        int sb = ph ? b1 : b2;
                        srcScore = S.fD[sb][i];
                        if(!ph) btD = mk_bt(S.fslot[sb][i], 0, K_DIAG, j);
                        else if(j < 128) btD = mk_bt(S.fslot[sb][i], 0, K_SGAP, j);
                        else btD = mk_bt(S.fslot[sb][i], 0, K_JUMP, j - 128);
                    } else if(dsel == 1) {
                        btD = mk_bt(slot, 1, K_HOP, -1);
                        int o = best_order(bG); int i = (o >> 8) & ((1 << C::IBITS) - 1); int j = o & 255;
                        srcScore = j ? S.fG[b1][i] : S.fD[b1][i];
                    } else {
                        btD = mk_bt(slot, 2, K_HOP, -1);
                        int o = best_order(bS); int i = (o >> 8) & ((1 << C::IBITS) - 1); int j = o & 255;
                        srcScore = (j & 1) ? S.fS[b1][i] : S.fD[b1][i];
                    }
                }
                int impMask = 0;
                int mD = Dv, mG = GGv, mS = SGv;          // merged values
                u64 mbtD = btD;                            // merged D back pointer
                if(pass == 0) {
                    // ---- new cells: write the table entry, register early / sequence-complete cells
                    if(isNew && S.err == 0 && slot < DP_CELLS) {
                        sl.cell_key[slot] = key;
                        sl.cell_sc[4 * slot + 0] = (short)Dv; sl.cell_sc[4 * slot + 1] = (short)GGv; sl.cell_sc[4 * slot + 2] = (short)SGv; sl.cell_sc[4 * slot + 3] = 0;
                        sl.cell_bt[slot] = btD; sl.cell_bt[DP_CELLS + slot] = btG; sl.cell_bt[2 * DP_CELLS + slot] = btS;
                    }
                    int x = key_x(key), y = key_y(key);
                    long long natural = (long long)(x > startLevel ? x - startLevel : startLevel - x) + (long long)(y > start_seq ? y - start_seq : start_seq - y);
                    bool isEarly = isNew && natural > d;
                    if(__ballot(isEarly)) {
                        if(!earlyInit) {
                            for(int i = lane; i < DP_EARLY; i += 64) sl.early_key[i] = HKEY_EMPTY;
                            earlyInit = true;
                            WSYNC();
                        }
                        if(isEarly) if(!early_insert(sl, key, slot)) S.err = __LINE__;
                    }
                    if(isNew && y == limitY) {                                                     // :982-999
                        int pos = atomicAdd(&S.nCompletedAdd, 1);
                        if(nCompleted + pos < DP_COMPLETED) sl.completed[nCompleted + pos] = slot; else S.err = __LINE__;
                    }
                }
                // ---- existing cells: each matrix independently overwritten iff strictly greater, :951-979 (writes are staged)
                if(keep && !isNew && S.err == 0) {
                    int oD = sl.cell_sc[4 * es + 0], oG = sl.cell_sc[4 * es + 1], oS = sl.cell_sc[4 * es + 2];
                    if(Dv > oD) impMask |= 1; else { mD = oD; mbtD = sl.cell_bt[es]; }
                    if(GGv > oG) impMask |= 2; else mG = oG;
                    if(SGv > oS) impMask |= 4; else mS = oS;
                    if(impMask && pass == 0) {
                        int p = atomicAdd(&S.nImp, 1);
                        if(p < DP_IMPCAP) {
                            sl.imp_slot[p] = es; sl.imp_key[p] = key; sl.imp_mask[p] = impMask;
                            sl.imp_new[4 * p + 0] = (short)mD; sl.imp_new[4 * p + 1] = (short)mG; sl.imp_new[4 * p + 2] = (short)mS;
                            sl.imp_bt[3 * p + 0] = btD; sl.imp_bt[3 * p + 1] = btG; sl.imp_bt[3 * p + 2] = btS;
                        } else S.err = __LINE__;
                    }
                }
                if(pass == 0 && slow) {
                    // exact diff needs every staged improvement of the iteration: finish in the second pass
                    if(act) { S.tes[t] = (short)slot; S.timp[t] = (unsigned char)(impMask | (isNew ? 0x80 : 0)); }
                    continue;
                }
                if(__ballot(impMask != 0)) anyOw = true;
                // ---- the `diff` rule, :1007-1041: score difference to the real previous step of the MERGED D back pointer
                int diff = 1;
                if(keep && S.err == 0) {
                    if(!slow) {
                        diff = Dv - srcScore;       // no table entry changes this iteration: cached frontier values are the table values
                    } else {
                        u64 b = mbtD;
                        int guard = 0;
                        while(bt_kind(b) == K_HOP && guard++ < 4) {
                            int m = bt_src(b);
                            bool useNew = isNew || (impMask & (1 << m));
                            if(useNew) b = (m == 1) ? btG : btS; else b = sl.cell_bt[m * DP_CELLS + slot];
                        }
                        int ps = bt_prev(b), pm = bt_src(b);
                        int pv = sl.cell_sc[4 * ps + pm];
                        // a predecessor improved in THIS iteration counts with its new value only if it precedes this cell in map order
                        int nImp = S.nImp < DP_IMPCAP ? S.nImp : DP_IMPCAP;
                        for(int q = 0; q < nImp; q++)
                            if(sl.imp_slot[q] == ps && (sl.imp_mask[q] & (1 << pm)) && sl.imp_key[q] < key) pv = sl.imp_new[4 * q + pm];
                        diff = Dv - pv;
                    }
                }
                // ---- running maximum bookkeeping, :1043-1062
                bool eq = keep && Dv == curMax0 && diff != 0;
                if(__ballot(eq)) anyEqDiff = true;
                int wm = wave_max_i32(keep ? Dv : DP_NEG);
                if(wm > itMaxNew) { itMaxNew = wm; itMaxKey = ~0ull; }
                u64 mn = wave_min_u64((keep && Dv == itMaxNew) ? key : ~0ull);
                if(mn < itMaxKey) itMaxKey = mn;
                // stash for the filter phase: [0] = slot (or ~0 if dropped), [1] = merged D | GG<<16, [2] = merged SG
                if(act) {
                    S.hbest[0][h] = keep ? (typename C::Best)(u32)slot : (typename C::Best)0xFFFFFFFFu;
                    S.hbest[1][h] = (typename C::Best)(((u32)(unsigned short)(short)mD) | ((u32)(unsigned short)(short)mG << 16));
                    S.hbest[2][h] = (typename C::Best)(u32)(unsigned short)(short)mS;
                }
            }
            WSYNC();
        }
        nCompleted += uni(S.nCompletedAdd);
        if(uni(S.err)) { DP_FAIL(__LINE__); break; }
        // apply staged improvements of existing cells and patch cached frontier copies
        {
            int nImp = uni(S.nImp);
            for(int q = lane; q < nImp; q += 64) {
                int es = sl.imp_slot[q]; int msk = sl.imp_mask[q];
                for(int m = 0; m < 3; m++) if(msk & (1 << m)) { sl.cell_sc[4 * es + m] = sl.imp_new[4 * q + m]; sl.cell_bt[m * DP_CELLS + es] = sl.imp_bt[3 * q + m]; }
            }
            if(nImp) {
                WSYNC();
                for(int q = 0; q < nImp; q++) {
                    int es = sl.imp_slot[q]; short v0 = sl.imp_new[4 * q + 0], v1 = sl.imp_new[4 * q + 1], v2 = sl.imp_new[4 * q + 2];
                    for(int i = lane; i < n1; i += 64) if(S.fslot[b1][i] == es) { S.fD[b1][i] = v0; S.fG[b1][i] = v1; S.fS[b1][i] = v2; }
                    for(int i = lane; i < n2; i += 64) if(S.fslot[b2][i] == es) { S.fD[b2][i] = v0; S.fG[b2][i] = v1; S.fS[b2][i] = v2; }
                }
            }
        }
        // "== currentMaximum && diff != 0" / "> currentMaximum" / overwritten entry all set lastMaximumIncrease_at_diagonalI
        if(itMaxNew > curMax0) {
            curMax = itMaxNew; lastInc = (int)d;
            int fs = -1;      // slot of the first cell in map order that carries the new maximum
            for(int t = lane; t < nT; t += 64) { int h = S.tlist[t]; if(S.hkey[h] == itMaxKey) fs = (int)S.hbest[0][h]; }
            firstMaxSlot = wave_max_i32(fs);
        }
        if(anyEqDiff || anyOw) lastInc = (int)d;

        if(timing) { long long t = clock64(); tEval += t - tMark; tMark = t; }
        // ================= filter + sort, :1076-1105 ======================================
        int mx = DP_NEG;
        for(int t = lane; t < nT; t += 64) { int h = S.tlist[t]; if((u32)S.hbest[0][h] != 0xFFFFFFFFu) { int v = (short)((u32)S.hbest[1][h] & 0xFFFF); mx = max(mx, v); } }
        mx = wave_max_i32(mx);
        int nNew = 0;
        for(int t0 = 0; t0 < nT; t0 += 64) {
            int t = t0 + lane;
            bool pass = false; u64 key = 0; int h = 0;
            if(t < nT) { h = S.tlist[t]; key = S.hkey[h]; if((u32)S.hbest[0][h] != 0xFFFFFFFFu) { int v = (short)((u32)S.hbest[1][h] & 0xFFFF); pass = (mx - v) <= 15; } }
            int rank = 0;
            if(pass) {
                for(int u = 0; u < nT; u++) {
                    int hu = S.tlist[u];
                    if((u32)S.hbest[0][hu] == 0xFFFFFFFFu) continue;
                    int vu = (short)((u32)S.hbest[1][hu] & 0xFFFF);
                    if((mx - vu) <= 15 && S.hkey[hu] < key) rank++;
                }
                if(rank < C::WCAP) {
                    S.fkey[bn][rank] = key; S.fslot[bn][rank] = (short)(int)S.hbest[0][h];
                    S.fD[bn][rank] = (short)((u32)S.hbest[1][h] & 0xFFFF); S.fG[bn][rank] = (short)((u32)S.hbest[1][h] >> 16); S.fS[bn][rank] = (short)((u32)S.hbest[2][h] & 0xFFFF);
                }
            }
            nNew += __popcll(__ballot(pass));
        }
        if(nNew > C::WCAP) { DP_FAIL(__LINE__); break; }
        WSYNC();
        // reset the hash entries used by this iteration
        for(int t = lane; t < nT; t += 64) { int h = S.tlist[t]; S.hkey[h] = HKEY_EMPTY; S.hbest[0][h] = 0; S.hbest[1][h] = 0; S.hbest[2][h] = 0; }
        if(lane == 0) S.nT = 0;
        WSYNC();
        { int tmp = b2; b2 = b1; b1 = bn; bn = tmp; }                                        // m2 := m1; m1 := this, :1104-1105
        if(timing) { long long t = clock64(); tFilt += t - tMark; tMark = t; }
        n2 = n1; n1 = nNew;
    }
    WSYNC();
    R.iters = (int)itersRun;
    R.cells = cellsEvaluated; R.edges = wave_sum_i32((int)edgesTouched);
    if(timing && lane == 0) { atomicAdd(&counters[8], (u64)tGen); atomicAdd(&counters[9], (u64)tEval); atomicAdd(&counters[10], (u64)tFilt); atomicAdd(&counters[11], (u64)itersRun); tMark = clock64(); }
    if(uni(S.err)) { R.err = uni(S.err); return R; }
    // ---- end cell, :1381-1517
    int endSlot = -1, endScore = 0;
    if(nCompleted > 0) {
        int best = DP_NEG;
        for(int i = lane; i < nCompleted; i += 64) best = max(best, (int)sl.cell_sc[4 * sl.completed[i] + 0]);
        best = wave_max_i32(best);
        int nTies = 0;
        for(int i0 = 0; i0 < nCompleted; i0 += 64) { int i = i0 + lane; bool tie = i < nCompleted && sl.cell_sc[4 * sl.completed[i] + 0] == best; nTies += __popcll(__ballot(tie)); }
        u32 sd = seed;
        int selectedIndex = glibc_rand_r(&sd) % nTies;                                      // Utilities.cpp:922-927
        // the tie with exactly `selectedIndex` ties before it in "x/z" string order
        int found = -1;
        for(int i0 = 0; i0 < nCompleted; i0 += 64) {
            int i = i0 + lane;
            if(i < nCompleted) {
                int s = sl.completed[i];
                if(sl.cell_sc[4 * s + 0] == best) {
                    u64 k = sl.cell_key[s]; int rank = 0;
                    if(nTies > 1) {
                        int kz = key_node(k) - G.level_off[key_x(k)];
                        for(int u = 0; u < nCompleted; u++) { int su = sl.completed[u]; if(su != s && sl.cell_sc[4 * su + 0] == best) { u64 ku = sl.cell_key[su];
                            if(xz_less(key_x(ku), key_node(ku) - G.level_off[key_x(ku)], key_x(k), kz)) rank++; } }
                    }
                    if(rank == selectedIndex) found = s;
                }
            }
        }
        endSlot = wave_max_i32(found); endScore = best;
    } else if(curMax > 0) {
        endSlot = firstMaxSlot; endScore = sl.cell_sc[4 * firstMaxSlot + 0];
    }
    if(endSlot < 0) return R;                                                                // no extension
    // ---- backtrace, :1109-1354: lane 0 chases the back pointers, then all lanes expand the steps into columns
    if(lane == 0) {
        int slot = endSlot, m = 0; u64 k = sl.cell_key[slot]; int x = key_x(k), y = key_y(k);
        int nSteps = 0, nCols = 0, guardSteps = 0;
        while((x != startLevel || y != start_seq) && nSteps < DP_STEPS && guardSteps++ < 4 * DP_STEPS) {
            u64 b = sl.cell_bt[m * DP_CELLS + slot];
            int kind = bt_kind(b);
            if(kind != K_HOP) {
                int len = 1;
                if(kind == K_JUMP) { int px = key_x(sl.cell_key[bt_prev(b)]); len = px > x ? px - x : x - px; }
                sl.step_bt[nSteps] = b; sl.step_xy[nSteps] = ((u64)(u32)x << 32) | ((u64)(u32)y << 8) | 0; nSteps++; nCols += len;
            }
            int prev = bt_prev(b);
            if(kind == K_DIAG) { x -= dir; y -= dir; }
            else if(kind == K_GGAP) { y -= dir; }
            else if(kind == K_SGAP) { x -= dir; }
            else if(kind == K_JUMP) { x = key_x(sl.cell_key[prev]); }
            slot = prev; m = bt_src(b);
        }
        S.nNew = nSteps; S.nKeepF = nCols;
        if(nSteps >= DP_STEPS || guardSteps >= 4 * DP_STEPS) S.err = __LINE__;
    }
    WSYNC();
    if(uni(S.err)) { R.err = uni(S.err); return R; }
    int nSteps = uni(S.nNew), nCols = uni(S.nKeepF);
    if(nCols > outCap) { R.err = -1000000 - nCols; return R; }
    int* oL = sl.x_level[side]; int* oE = sl.x_edge[side]; uint8_t* oG = sl.x_g[side]; uint8_t* oS = sl.x_s[side];
    int base = 0;
    for(int s0 = 0; s0 < nSteps; s0 += 64) {
        int s = s0 + lane; bool act = s < nSteps;
        u64 b = act ? sl.step_bt[s] : 0; u64 xy = act ? sl.step_xy[s] : 0;
        int kind = bt_kind(b);
        // resolve the graph object behind the push index j: edge j of the previous cell's node, or entry j of its jump table
        int pnode = 0, plevel = 0, robj = -1;
        if(act && kind != K_GGAP) {
            u64 pkey = sl.cell_key[bt_prev(b)]; pnode = key_node(pkey); plevel = key_x(pkey);
            int j = bt_edge(b);
            if(kind == K_JUMP) robj = (fwd ? G.jf_path : G.jb_path)[(fwd ? G.jf_off : G.jb_off)[pnode] + j];
            else robj = fwd ? G.out_eid[G.out_off[pnode] + j] : G.in_eid[G.in_off[pnode] + j];
        }
        int len = act ? (kind == K_JUMP ? G.path_len[robj] : 1) : 0;
        int total; int off = wave_excl_scan(len, total);
        if(act) {
            int x = (int)(xy >> 32), y = (int)((xy >> 8) & 0xFFFFFF);
            int start = fwd ? (nCols - (base + off) - len) : (base + off);     // forward traces are reversed at the end, :1319-1326
            if(kind == K_JUMP) {                                                       // :1282-1307
                int p = robj; long long po = G.path_off[p];
                int lvl0 = G.node_level[G.edge_from_new[G.path_edges[po]]];
                for(int j = 0; j < len; j++) { oL[start + j] = lvl0 + j; oE[start + j] = G.path_edges[po + j]; oG[start + j] = '_'; oS[start + j] = '_'; }
            } else {
                int eid = robj;
                unsigned char sc = fwd ? (y >= 1 ? S.seq[y - 1] : 0) : (y < max_seqI ? S.seq[y] : 0);
                int lvl = fwd ? x - 1 : x;
                if(kind == K_DIAG) { oL[start] = lvl; oE[start] = eid; oG[start] = G.edge_label[eid]; oS[start] = sc; }
                else if(kind == K_GGAP) { oL[start] = -1; oE[start] = -1; oG[start] = '_'; oS[start] = sc; }
                else { oL[start] = lvl; oE[start] = eid; oG[start] = G.edge_label[eid]; oS[start] = '_'; }
            }
        }
        base += total;
    }
    WSYNC();
    if(timing && lane == 0) { atomicAdd(&counters[12], (u64)(clock64() - tMark)); }
    u64 ek = sl.cell_key[endSlot];
    int yEnd = key_y(ek);
    R.have = 1; R.ncols = nCols; R.score = endScore;
    if(fwd) { R.seq_begin = start_seq; R.seq_end = yEnd - 1; }                               // toVerboseSeedChain, VirtualNWUnique.cpp:28-29
    else { R.seq_begin = yEnd; R.seq_end = start_seq - 1; }
    if(R.seq_begin > R.seq_end) { R.err = __LINE__; R.have = 0; }
    return R;
}

// ------------------------------------------------------------------------------------------
// one wave per chain: left DP, right DP, stitch (extendWithOtherSeedChain / extendToFullSequenceLength,
// verboseSeedChain.cpp:23-136), then scoreOneAlignment (extensionAligner.cpp:52-182).
template <class C, bool RETRY>
__global__ __launch_bounds__(64, 5) void k_extend_chains(const DevGraph* __restrict__ Gp, const DevTables* __restrict__ Tp, const DevBatch* __restrict__ Bp, char* slabs, size_t slabBytes, u32 rng_seed)
{
    // graph / batch descriptors stay in memory (scalar loads on demand): passing them by value costs ~150 SGPRs
    const DevGraph& G = *Gp;
    const DevBatch& B = *Bp;
    __shared__ DpLdsT<C> S;
    const int lane = lane_id();
    const DevTables& T = *Tp;
    ExtSlab sl = ext_slab_at(slabs + (size_t)blockIdx.x * slabBytes, B.stride);
    const int stride = B.stride;

    for(;;) {
        int c;
        if(RETRY) { int i = next_work(&B.work_counter[4]); if(i >= uni(B.work_counter[3])) break; c = uni(B.retry_list[i]); }
        else { c = next_work(&B.work_counter[1]); if(c >= B.n_chains) break; }
        int st = uni(B.seed_status[c]);
        long long tChain0 = B.dbg ? clock64() : 0;
        if(st != HLALA_CHAIN_OK) {
            if(lane == 0 && !RETRY) { B.ext_status[c] = st; B.ext_ncols[c] = 0; B.dp_iters[2 * c] = 0; B.dp_iters[2 * c + 1] = 0; B.dp_score[2 * c] = INT32_MIN; B.dp_score[2 * c + 1] = INT32_MIN;
                            if(st < 0) atomicAdd(&B.counters[CNT_ERRORS], 1ull); }
        } else {
        const int r = uni(B.chain_read[c]);
        const int rOff = uni(B.read_off[r]), seqLen = uni(B.read_off[r + 1]) - rOff;
        const size_t cb = (size_t)c * stride;
        const int nSeed = uni(B.seed_ncols[c]), sBegin = uni(B.seed_begin[c]), sEnd = uni(B.seed_end[c]);
        int err = 0;
        if(seqLen > C::SEQCAP && !RETRY && seqLen <= DP_SEQCAP && nSeed >= 1) err = HLALA_CHAIN_ERR_FRONTIER;      // long read: large class
        else if(seqLen > C::SEQCAP || seqLen < 1 || nSeed < 1 || sBegin < 0 || sEnd >= seqLen || sBegin > sEnd) err = HLALA_CHAIN_ERR_INPUT;
        if(!err) for(int i = lane; i < seqLen; i += 64) S.seq[i] = B.read_bases[rOff + i];
        WSYNC();
        DpResult RL, RR; RL.have = 0; RL.ncols = 0; RL.iters = 0; RL.score = INT32_MIN; RL.err = 0; RL.seq_begin = 0; RL.seq_end = -1; RL.cells = 0; RL.edges = 0; RR = RL;
        int nCalls = 0;
        if(!err) {
            int e0 = uni(B.seed_edge[cb]), e1 = uni(B.seed_edge[cb + nSeed - 1]);
            if(e0 < 0 || e1 < 0 || e0 >= G.E || e1 >= G.E) err = HLALA_CHAIN_ERR_INPUT;
            else {
                if(sBegin != 0) {                                                      // left extension, extensionAligner.cpp:220-268
                    int firstNode = uni(G.edge_from_new[e0]); int lvl = uni(G.node_level[firstNode]);
                    if(lvl > 0) { nCalls++; RL = dp_run<C>(G, S, sl, seqLen, sBegin, lvl, firstNode, false, rng_seed + 2u * (u32)c, 0, stride, B.counters, B.dbg, c); }
                }
                if(sEnd != seqLen - 1) {                                               // right extension, :271-319
                    int lastNode = uni(G.edge_to_new[e1]); int lvl = uni(G.node_level[lastNode]);
                    if(lvl < G.L - 1) { nCalls++; RR = dp_run<C>(G, S, sl, seqLen, sEnd + 1, lvl, lastNode, true, rng_seed + 2u * (u32)c + 1u, 1, stride, B.counters, B.dbg, c); }
                }
                if(RL.err || RR.err) err = ((RL.err <= -1000000) || (RR.err <= -1000000)) ? HLALA_CHAIN_ERR_COLUMNS : HLALA_CHAIN_ERR_FRONTIER;
            }
        }
        int nL = RL.have ? RL.ncols : 0, nR = RR.have ? RR.ncols : 0;
        int newBegin = RL.have ? RL.seq_begin : sBegin, newEnd = RR.have ? RR.seq_end : sEnd;
        int padL = newBegin, padR = seqLen - 1 - newEnd;
        int total = padL + nL + nSeed + nR + padR;
        if(!err && total > stride) err = HLALA_CHAIN_ERR_COLUMNS;
        if(lane == 0) {
            B.dp_iters[2 * c] = RL.iters; B.dp_iters[2 * c + 1] = RR.iters;
            B.dp_score[2 * c] = RL.have ? RL.score : INT32_MIN; B.dp_score[2 * c + 1] = RR.have ? RR.score : INT32_MIN;
        }
        // a chain that outgrew the small capacity class is queued for the large-capacity pass and leaves no trace here
        const bool requeue = !RETRY && err == HLALA_CHAIN_ERR_FRONTIER;
        if(lane == 0 && !requeue) {
            atomicAdd(&B.counters[CNT_DP_CALLS], (u64)nCalls); atomicAdd(&B.counters[CNT_DP_ITERS], (u64)(RL.iters + RR.iters));
            atomicAdd(&B.counters[CNT_DP_CELLS], RL.cells + RR.cells); atomicAdd(&B.counters[CNT_EDGES], (u64)(RL.edges + RR.edges));
        }
        if(err) {
            if(lane == 0) {
                if(requeue) { int q = atomicAdd(&B.work_counter[3], 1); B.retry_list[q] = c; }
                else { B.ext_status[c] = err; B.ext_ncols[c] = 0; B.ext_ll[c] = (double)(RL.err ? RL.err : RR.err); atomicAdd(&B.counters[CNT_ERRORS], 1ull); }
            }
        } else {
        long long tStitch0 = B.dbg ? clock64() : 0;
        // ---- stitch
        for(int j = lane; j < total; j += 64) {
            int lvl, edge; unsigned char g, s, fs = 0;
            if(j < padL) { lvl = -1; edge = -1; g = '_'; s = S.seq[j]; }
            else if(j < padL + nL) { int q = j - padL; lvl = sl.x_level[0][q]; edge = sl.x_edge[0][q]; g = sl.x_g[0][q]; s = sl.x_s[0][q]; }
            else if(j < padL + nL + nSeed) { int q = j - padL - nL; lvl = B.seed_level[cb + q]; edge = B.seed_edge[cb + q]; g = B.seed_g[cb + q]; s = B.seed_s[cb + q]; fs = 1; }
            else if(j < padL + nL + nSeed + nR) { int q = j - padL - nL - nSeed; lvl = sl.x_level[1][q]; edge = sl.x_edge[1][q]; g = sl.x_g[1][q]; s = sl.x_s[1][q]; }
            else { int q = j - (padL + nL + nSeed + nR); lvl = -1; edge = -1; g = '_'; s = S.seq[newEnd + 1 + q]; }
            B.ext_level[cb + j] = lvl; B.ext_edge[cb + j] = edge; B.ext_g[cb + j] = g; B.ext_s[cb + j] = s; B.ext_fromseed[cb + j] = fs;
        }
        WSYNC();
        // ---- scoreOneAlignment: terms are added strictly left to right in FP64 (same order as the reference loop).
        // The read quality of alignment-orientation base i is read_quals[i] in both strands: the reference indexes the
        // ORIGINAL-orientation read with len-i-1 when the chain is reverse (extensionAligner.cpp:85-88), which is the same base.
        {
            // phase 1 (parallel): every lane turns its <= 8 consecutive columns into two addends each (an unused addend is +0.0,
            // an exact identity); phase 2 (serial by construction of FP addition): the running sum walks the lanes in order.
            constexpr int LLPER = 8;                                   // 64 * 8 = 512 >= params.max_columns
            const int per = (total + 63) / 64;
            const int j0 = lane * per, j1 = min(total, j0 + per);
            unsigned char scv[LLPER], gcv[LLPER];
            int nb = 0;
#pragma unroll
            for(int k = 0; k < LLPER; k++) {
                int j = j0 + k; bool in = k < per && j < j1;
                scv[k] = in ? B.ext_s[cb + j] : (unsigned char)'_'; gcv[k] = in ? B.ext_g[cb + j] : (unsigned char)'_';
                if(in && scv[k] != '_') nb++;
            }
            int tot; int before = wave_excl_scan(nb, tot);
            double t1[LLPER], t2[LLPER];
            {
                int idx = before;
#pragma unroll
                for(int k = 0; k < LLPER; k++) {
                    unsigned char sc = scv[k], gc = gcv[k];
                    double a1 = 0.0, a2 = 0.0;
                    if(sc != '_') {
                        if(gc == '_') a1 = T.rate_ins_quarter;
                        else { a1 = T.rate_match_mismatch; unsigned char q = B.read_quals[rOff + idx]; a2 = (sc == gc) ? T.ll_match[q] : T.ll_mismatch[q]; }
                        idx++;
                    } else if(gc != '_') a1 = T.rate_indel;
                    t1[k] = a1; t2[k] = a2;
                }
            }
            double acc = 0.0;
            for(int l = 0; l < 64; l++) {
                double in = __shfl(acc, l > 0 ? l - 1 : 0);
                if(lane == l) {
                    double a = (l == 0) ? 0.0 : in;
#pragma unroll
                    for(int k = 0; k < LLPER; k++) { a += t1[k]; a += t2[k]; }
                    acc = a;
                }
            }
            double ll = __shfl(acc, 63);
            // first / last two defined levels for the pairing stage (verboseSeedChain.h:134-228)
            if(lane == 0) {
                int f0 = -1, f1 = -1, l0 = -1, l1 = -1;
                for(int j = 0; j < total && f1 < 0; j++) { int lv = B.ext_level[cb + j]; if(lv != -1) { if(f0 < 0) f0 = lv; else f1 = lv; } }
                for(int j = total - 1; j >= 0 && l1 < 0; j--) { int lv = B.ext_level[cb + j]; if(lv != -1) { if(l0 < 0) l0 = lv; else l1 = lv; } }
                B.ext_firstlast[4 * c + 0] = f0; B.ext_firstlast[4 * c + 1] = f1; B.ext_firstlast[4 * c + 2] = l0; B.ext_firstlast[4 * c + 3] = l1;
                B.ext_status[c] = HLALA_CHAIN_OK; B.ext_ncols[c] = total; B.ext_begin[c] = 0; B.ext_end[c] = seqLen - 1; B.ext_ll[c] = ll;
                atomicAdd(&B.counters[CNT_CHAINS_EXT], 1ull); atomicAdd(&B.counters[CNT_OUT_COLS], (u64)total);
                if(B.dbg) { long long t = clock64(); atomicAdd(&B.counters[13], (u64)(t - tStitch0)); atomicAdd(&B.counters[14], (u64)(t - tChain0)); atomicAdd(&B.counters[15], 1ull); }
            }
        }
        }   // no error
        }   // chain to extend
        WSYNC();
    }
}

}  // namespace hlala
