// kernel_project.hip -- stage A: BAM record -> graph-space seed chain (processBAM::alignment2Chain) on gfx950.
//
//   k_filter_chains  : strand / duplicate-coordinate filters of alignOneReadPair (processBAM.cpp:3200-3240),
//                      one thread per read (the filter is sequential over a read's chains by definition).
//   k_project_chains : one wavefront per surviving chain:
//        CIGAR walk -> columns            transformBAMreadToInternalAlignment   processBAM.cpp:4794-5337
//        trim / pad skipped levels        PRGContigAlignment2Seed               :2518-2579
//        cleanInitialAlignment                                                  :4621-4792
//        restrictInitialAlignmentToNoGapAreas                                   :4461-4619
//        re-threading DP (sequence variant) + deterministic backtrace           :2676-3007
//      Columns live in LDS.  The chain's window of the in-edge CSR (level offsets, edge offsets, from-nodes,
//      labels) is staged into LDS once and the per-node back pointers stay there too (the common case);
//      windows that exceed PROJ_SN nodes / PROJ_SE edges run the same recurrence against HBM with the
//      back pointers in the wave's slab.
#include "batch.h"
#include "../../include/hlala_gpu.h"

namespace hlala {

constexpr int PROJ_CAP   = 512;     // alignment columns held in LDS (>= params.max_columns)
constexpr int PROJ_NODES = 512;     // nodes per level held in LDS score rows
constexpr int PROJ_SN    = 640;     // nodes of a chain's level window staged in LDS (CSR offsets + back pointers)
constexpr int PROJ_SE    = 768;     // in-edges of that window staged in LDS
constexpr int PROJ_SEGMAX = 24;     // longest run of multi-node levels one lane solves alone

struct ChoiceRec { int eid; short fromz; short S; };

// k_rethread_chains (below): the chunked form of the re-threading DP in a kernel of its own, with a fraction of the LDS and registers of k_project_chains
constexpr int RT_SN = 416;          // nodes of a chunk of levels (and of one level)
constexpr int RT_CE = 512;          // in-edges of a chunk (and of one level)
constexpr int CHAIN_RETHREAD_PENDING = 64;     // seed_status between the two kernels (never leaves the stage)


// (CAP_ = columns held; two sizes are instantiated: 512 and -- for params.max_columns <= 384, the default of the paired path -- 384, whose
// 14.5 KB let 11 waves share a CU's LDS instead of 9: the kernel waits on memory two thirds of its cycles)
template <int CAP_, int SN_, int SE_>
struct __align__(16) ProjLdsT {
    static constexpr int CAP = CAP_, SN = SN_, SE = SE_; static constexpr bool LONG = false;
    typedef unsigned short LvT;             // a column's level relative to the chain's first level, + 1; 0 = none (LvCodec)
    LvT lvl[2][CAP_];
    unsigned char g[2][CAP_], s[2][CAP_];
    short Srow[2][PROJ_NODES];
    // the chain's window of the in-edge CSR, staged once so that the per-column recurrence never leaves LDS
    unsigned short sLev[CAP_ + 2];          // level_off[level0 + i] - nodeBase
    unsigned short sIn[SN_ + 1];            // in_off[tgtBase + i] - eBase
    unsigned short sChoice[SN_];            // per target node: chosen in-edge (index into the window), 0xFFFF = unreachable
    static constexpr int CE = (3 * SE_) / 4;        // in-edges of a chunk of the chunked form (its packed records share the bytes of sFrom and sLab)
    union {
        struct { unsigned short sFrom[SE_];     // in_from[eBase + e] - nodeBase
                 unsigned char sLab[SE_]; };
        u32 eRec[CE];                       // chunked form: DevGraph::in_rec of the chunk's in-edges
    };
    u32 colInfo[CAP_];                      // per level of the window: column | read char << 16 | seed-is-match << 24
    unsigned short segStart[CAP_ + 2];      // level indices where a DP segment starts (single-node levels, see below)
    u64 mGap[CAP_ / 64], mDef[CAP_ / 64], mSeq[CAP_ / 64];     // column bit masks of the restrict step
    int err, n, startRaw, stopRaw, tmp0, tmp1;
    __device__ __forceinline__ short* sflat() { return &Srow[0][0]; }        // S per node of the window in the segment-parallel form (SN <= 2 * PROJ_NODES)
};
typedef ProjLdsT<PROJ_CAP, PROJ_SN, PROJ_SE> ProjLds;
// the layout of the paired path (params.max_columns <= 384): a 2x150 bp chain spans ~170-200 levels, i.e. ~280 nodes / ~300 in-edges on backbone
// stretches; 416 / 544 keep those in the one-shot staged form and, with 16-bit column levels, make the block 11.4 KB: 13 blocks per CU by LDS (12 resident: 143 VGPRs, three waves per SIMD)
constexpr int PROJ_CAP_SHORT = 384;
typedef ProjLdsT<PROJ_CAP_SHORT, 416, 544> ProjLdsShort;

// Long reads (params.max_columns > PROJ_CAP): the same kernel with the column / window arrays in the wave's HBM slab; the LDS block
// only holds the pointers, the per-level score rows and the scalars.  Member names and index syntax match ProjLds.
constexpr int PROJL_CAP = 16384;    // alignment columns (>= params.max_columns)
constexpr int PROJL_SN  = 49152;    // nodes of the level window (16-bit offsets: < 65536)
constexpr int PROJL_SE  = 57344;    // in-edges of the level window
constexpr int PROJL_LONGSEG = 32;   // long segments (more than PROJ_SEGMAX levels without a cut) of one read that are solved level by level beside its short ones; a read with more takes that form as a whole
struct __align__(16) ProjLdsLong {
    static constexpr int CAP = PROJL_CAP, SN = PROJL_SN, SE = PROJL_SE; static constexpr bool LONG = true;
    typedef int LvT;                        // absolute levels, -1 = none
    static constexpr int CE = PROJL_SE;
    GPtr<int> lvl[2];                       // (GPtr: device_common.h -- accesses are global_load / global_store, not flat)
    GPtr<unsigned char> g[2], s[2];
    GPtr<unsigned short> sLev, sIn, sChoice, sFrom; GPtr<unsigned char> sLab;
    GPtr<u32> colInfo; GPtr<unsigned short> segStart;
    GPtr<unsigned long long> chunkStart;     // [CAP / 64] bitmap of the levels at which a chunk of the level-by-level form started: bits set with atomics, read with atomic loads (the L1 may hold the last read's words)
    union {
        struct { u64 mGap[PROJL_CAP / 64], mDef[PROJL_CAP / 64], mSeq[PROJL_CAP / 64]; };      // column bit masks of the clean / restrict steps: in LDS (6 KB) -- the scalar walks over them are chains of dependent reads (round 6: they were in the slab)
        // the re-threading DP (the masks are dead by then): the score rows of the level loops, a chunk's in-edge records and in-edge offsets -- as in k_rethread_chains --, and the
        // level ranges that go through the level-by-level form (the long segments of a read, or the whole read).  6.3 KB of LDS per block in all: 20 blocks per CU at five per SIMD.
        struct { short Srow[2][PROJ_NODES]; u32 cRec[RT_CE]; unsigned short cIn[RT_SN + 2]; int longSeg[2 * PROJL_LONGSEG]; };
    };
    short* sflatp;
    int err, n, startRaw, stopRaw, tmp0, tmp1;
    __device__ __forceinline__ HLALA_AS_GLOBAL short* sflat() { return glob(sflatp); }
};
__host__ __device__ inline size_t proj_long_slab_bytes()
{
    size_t b = 0;
    b += 2 * (size_t)PROJL_CAP * 4 + 4 * (size_t)PROJL_CAP;                                   // lvl, g, s
    b += ((size_t)PROJL_CAP + 2) * 2 * 2 + (size_t)PROJL_CAP * 4 + 3 * (size_t)(PROJL_CAP / 64) * 8;     // sLev, segStart, colInfo, masks
    b += ((size_t)PROJL_SN + 2) * 2 + 2 * (size_t)PROJL_SN * 2 + (size_t)PROJL_SE * 2 + (size_t)PROJL_SE;    // sIn, sChoice, sflat, sFrom, sLab
    return (b + 4095) & ~(size_t)255;
}
template <int CAP_, int SN_, int SE_> __device__ inline void proj_bind(ProjLdsT<CAP_, SN_, SE_>&, char*) { }
__device__ inline void proj_bind(ProjLdsLong& P, char* p)      // 8-byte arrays first, then 4-, 2-, 1-byte ones
{
    P.chunkStart = (unsigned long long*)p; p += 3 * (size_t)(PROJL_CAP / 64) * 8;       // (the masks lived here until round 6: the first third is the chunk bitmap now)
    P.lvl[0] = (int*)p; p += (size_t)PROJL_CAP * 4; P.lvl[1] = (int*)p; p += (size_t)PROJL_CAP * 4; P.colInfo = (u32*)p; p += (size_t)PROJL_CAP * 4;
    P.sLev = (unsigned short*)p; p += ((size_t)PROJL_CAP + 2) * 2; P.segStart = (unsigned short*)p; p += ((size_t)PROJL_CAP + 2) * 2;
    P.sIn = (unsigned short*)p; p += ((size_t)PROJL_SN + 2) * 2; P.sChoice = (unsigned short*)p; p += (size_t)PROJL_SN * 2; P.sflatp = (short*)p; p += (size_t)PROJL_SN * 2; P.sFrom = (unsigned short*)p; p += (size_t)PROJL_SE * 2;
    P.g[0] = (unsigned char*)p; p += PROJL_CAP; P.g[1] = (unsigned char*)p; p += PROJL_CAP; P.s[0] = (unsigned char*)p; p += PROJL_CAP; P.s[1] = (unsigned char*)p; p += PROJL_CAP;
    P.sLab = (unsigned char*)p;
}

__host__ __device__ inline size_t proj_slab_bytes(int stride, int maxNodesPerLevel)
{
    size_t ent = (size_t)stride * (size_t)(maxNodesPerLevel < 1 ? 1 : maxNodesPerLevel);
    if(ent > (1u << 20)) ent = (1u << 20);
    if(ent < 1024) ent = 1024;
    return (ent * sizeof(ChoiceRec) + 255) & ~(size_t)255;
}

// Column levels in LDS are 16 bits: level - (first level of the chain) + 1, 0 = none (levels never decrease along a chain and a chain spans fewer
// levels than it has column slots, so anything that does not fit is a chain the int form flags as well).  The long-read layout keeps ints in HBM.
// The other column buffer doubles as the per-column edge pick of the backtrace: window-relative edge index, all ones = none.
template <class PL> struct LvCodec {
    typedef typename PL::LvT T;
    static constexpr bool SHORT = sizeof(T) == 2;
    __device__ static __forceinline__ int get(T v, int base) { if constexpr (SHORT) return v ? base + (int)v - 1 : -1; else return (int)v; }
    __device__ static __forceinline__ T put(int lv, int base) { if constexpr (SHORT) return (T)(lv < 0 ? 0 : lv - base + 1); else return (T)lv; }
    __device__ static __forceinline__ T pickNone() { if constexpr (SHORT) return (T)0xFFFF; else return (T)-1; }
    __device__ static __forceinline__ bool pickIsNone(T v) { if constexpr (SHORT) return v == (T)0xFFFF; else return (int)v < 0; }
};

// BamAlignment::GetEndPosition(false, true) of BamTools 2.5.1 (un-vendored dependency, makefile:3-12):
// Position + sum of M, =, X, D, N lengths - 1.  Call site processBAM.cpp:3872.
__device__ inline int cigar_end_position(const u32* cig, int n, int pos)
{
    int e = pos;
    for(int i = 0; i < n; i++) { u32 op = cig[i] & 15u; if(op == 0 || op == 7 || op == 8 || op == 2 || op == 3) e += (int)(cig[i] >> 4); }
    return e - 1;
}

__global__ void k_filter_chains(const DevGraph* __restrict__ Gp, const DevBatch* __restrict__ Bp, const long long* contig_off, const int* contig_level)
{
    const DevBatch& B = *Bp;
    int r = blockIdx.x * blockDim.x + threadIdx.x;
    if(r >= B.n_reads) return;
    int c0 = B.chain_off[r], c1 = B.chain_off[r + 1];
    bool primRev = B.chain_reverse[B.read_primary[r]] != 0;
    for(int c = c0; c < c1; c++) {
        int st = HLALA_CHAIN_OK;
        int contig = B.chain_contig[c];
        long long clen = contig_off[contig + 1] - contig_off[contig];
        int a = B.chain_pos[c] - B.chain_offset[c];
        int b = cigar_end_position(B.cigar + B.cigar_off[c], B.cigar_off[c + 1] - B.cigar_off[c], B.chain_pos[c]) - B.chain_offset[c];
        int idA = -1, idB = -1;
        if(a < 0 || a >= clen || b < 0 || b >= clen) st = HLALA_CHAIN_ERR_INPUT;          // asserts of alignment_get_startstop_PRGcoordinates, :3883-3885
        else { idA = contig_level[contig_off[contig] + a]; idB = contig_level[contig_off[contig] + b]; }
        if(st == HLALA_CHAIN_OK && (B.chain_reverse[c] != 0) != primRev) st = HLALA_CHAIN_SKIP_STRAND;      // :3216
        if(st == HLALA_CHAIN_OK) {
            // "start//stop" id already extended with >= AS (:3207, :3234); the map holds the best score per id
            for(int p = c0; p < c; p++) {
                if(B.seed_status[p] != HLALA_CHAIN_OK) continue;
                int pc = B.chain_contig[p];
                int pa = B.chain_pos[p] - B.chain_offset[p];
                int pb = cigar_end_position(B.cigar + B.cigar_off[p], B.cigar_off[p + 1] - B.cigar_off[p], B.chain_pos[p]) - B.chain_offset[p];
                int pidA = contig_level[contig_off[pc] + pa], pidB = contig_level[contig_off[pc] + pb];
                if(pidA == idA && pidB == idB && B.chain_as[p] >= B.chain_as[c]) { st = HLALA_CHAIN_SKIP_DUP; break; }
            }
        }
        B.seed_status[c] = st;
        if(st != HLALA_CHAIN_OK) B.seed_ncols[c] = 0;
        // position bucket of the chain (kernel_order.hip): the level its first reference base sits on; chains that take no further part go last
        // (one atomic per chain on ONE counter -- all filtered chains in a bucket of their own -- cost 90 ms per million pairs: same-address atomics
        //  serialise at the L2)
        if(B.chain_bucket) {
            int bk = -1;
            if(st == HLALA_CHAIN_OK) {
                if(B.order_cost) {
                    // long reads: heaviest window first (32 nodes per bucket; a 10 kb backbone read has ~11 k nodes in its window, one across a gene window 100 k and more)
                    const int la = idA < idB ? idA : idB, lb = idA < idB ? idB : idA;
                    const int cost = (Gp->level_off[lb + 1] - Gp->level_off[la]) >> 5;
                    bk = B.order_nb - 2 - cost; if(bk < 0) bk = 0;
                } else { bk = (idA >= 0 ? idA : (idB >= 0 ? idB : 0)) >> B.order_shift; if(bk > B.order_nb - 2) bk = B.order_nb - 2; }
                atomicAdd(&B.order_hist[bk], 1);
            }
            B.chain_bucket[c] = bk;
        }
    }
}

// Copy loops global -> LDS (staging a window of the graph): `for(i ...) lds[i] = f(global[i])` waits for every load before its store, one
// round trip per 64 (or 16) elements.  Here U loads are requested before the first store: one round trip per U rows of lanes.
template <int U, class LoadF, class StoreF>
__device__ __forceinline__ void staged_rows(int first, int end, int step, LoadF load, StoreF store)
{
    for(int base = first; base < end; base += U * step) {
        decltype(load(0)) v[U];
        #pragma unroll
        for(int u = 0; u < U; u++) { const int i = base + u * step; if(i < end) v[u] = load(i); }
        #pragma unroll
        for(int u = 0; u < U; u++) { const int i = base + u * step; if(i < end) store(i, v[u]); }
    }
}
struct FromLab { int from; unsigned char lab; };

// colInfo, chunked form: bits 25-29 = DevGraph::level_fast of the level (25-26: 1 = edge-parallel in one slice of 64 in-edges, 2 = in several, 0 = node by node; 27-29: largest in-degree - 1)
#define PJ_FAIL(code) do { if(P.err == 0) P.err = (code); } while(0)
#ifdef HLALA_PROJ_TIMING       // build-time switch (make EXTRA=-DHLALA_PROJ_TIMING; tools/long_phase.py): the long-read projection in pieces, per-read histograms, the heaviest reads on their own
constexpr bool PJ_FINE = true;
#else
constexpr bool PJ_FINE = false;
#endif
#define PJ_T(i) do { if(B.dbg) tPh[i] = clock64(); } while(0)      // HLALA_DEBUG phase clocks -> counters[16..23]
// Ordering between the lanes of the wavefront for LDS ONLY: the LDS instructions of a wavefront execute in order, so waiting for the LDS counter and keeping the compiler from
// moving accesses across is enough.  WSYNC() is a release / acquire fence and also waits for every store to HBM in flight -- a microsecond -- which the level loops of
// the long-read layout paid once per level (their stores of back pointers into the slab are not read before the backtrace): 10 000 levels, 5 k cycles each (round 6).
#define LSYNC() do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_wave_barrier(); } while(0)
#define PJ_OK() (uni(P.err) == 0)      // read at points where every lane has passed a barrier: wave-uniform

constexpr int PJL_U = 4;            // rows of lanes the copy loops of the long-read instantiation keep in flight (registers: the kernel is compiled for five wavefronts per SIMD)
#ifndef HLALA_PROJ_LONG_WPS
#define HLALA_PROJ_LONG_WPS 5      // wavefronts per SIMD the long-read instantiation is compiled for (5: 96 registers, 3 spilled, 20 blocks of 6.3 KB of LDS per CU; 4: 123 registers, 16 blocks -- 50.0 against 43.5 ms per 50 000 reads)
#endif
#ifndef HLALA_PROJ_WPS
#define HLALA_PROJ_WPS 2           // the LDS layouts: the compiler's own choice (153 registers = three wavefronts per SIMD; their 11.7 KB of LDS allow 13 blocks per CU)
#endif
template <class PL>
__global__ __launch_bounds__(64, PL::LONG ? HLALA_PROJ_LONG_WPS : HLALA_PROJ_WPS) void k_project_chains(const DevGraph* __restrict__ Gp, const DevBatch* __restrict__ Bp, const long long* contig_off, const uint8_t* contig_seq,
                                                       const int* contig_level, char* slabs, size_t slabBytes, char* longSlabs, size_t longSlabBytes,
                                                       int deferRethread)      // 1: chains that need the chunked form and fit k_rethread_chains are left to it
{
    const DevGraph& G = *Gp;
    const DevBatch& B = *Bp;
    __shared__ PL P;
    const int lane = lane_id();
    if(PL::LONG) { if(lane == 0) proj_bind(P, longSlabs + (size_t)blockIdx.x * longSlabBytes); WSYNC(); }
    if(PL::LONG && deferRethread > 0) { const long long tGo = clock64() + (long long)(blockIdx.x & 63) * (long long)deferRethread; while(clock64() < tGo) __builtin_amdgcn_s_sleep(64); }      // (experiment: staggered starts)
    ChoiceRec* slabCh = (ChoiceRec*)(slabs + (size_t)blockIdx.x * slabBytes);
    const int slabEnt = (int)(slabBytes / sizeof(ChoiceRec));
    const int stride = B.stride;

    long long tAcc[7] = {0, 0, 0, 0, 0, 0, 0};
    int readOrd = -1;                          // HLALA_DEBUG: how many chains this wavefront has finished
    long long tSub[4] = {0, 0, 0, 0};          // HLALA_DEBUG: chunked form -- chunk staging, level loops, chunks, levels
    long long tFine[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tFineLast = 0;      // HLALA_DEBUG: pieces of the re-threading DP and of the backtrace (dbg[4300 ..], cycles >> 12; tools/long_phase.py)
#define PJ_F(i) do { if constexpr (PL::LONG && PJ_FINE) if(B.dbg) { const long long t_ = clock64(); if((i) >= 0) tFine[(i) < 0 ? 0 : (i)] += t_ - tFineLast; tFineLast = t_; } } while(0)
    u64 accCols = 0, accEdges = 0;           // work counters, flushed once per wave (same-address atomics serialise at the L2)
    constexpr int CHUNK = PL::LONG ? 1 : 4;  // chains drawn per atomic (long reads, heaviest first: one -- a draw of four gave the first wavefronts the four heaviest reads of the batch each)
    const int nWork = ordered_chains(B);     // position order: only the chains that passed the filters are listed
    for(;;) {
        int c0 = 0;
        if(lane == 0) c0 = atomicAdd(&B.work_counter[0], CHUNK);
        c0 = __builtin_amdgcn_readfirstlane(c0);
        if(c0 >= nWork) break;
        const int cEnd = min(c0 + CHUNK, nWork);
        // the descriptors of the chunk's chains: lane q holds chain c0 + q; two round trips for the chunk (the fields, then what they point to)
        // instead of a chain of dependent wave-uniform loads per chain
        // (chains are taken in the order of their graph position, B.chain_order -- kernel_order.hip: the windows of the chains in flight at one time are
        //  neighbours in the CSR arrays and stay in the L2)
        int hStatus = -1, hRead = 0, hContig = 0, hPos = 0, hOff = 0, hCg0 = 0, hCg1 = 0, hChain = 0;
        if(lane < CHUNK && c0 + lane < cEnd) {
            const int cq = B.chain_order ? B.chain_order[c0 + lane] : c0 + lane;
            hChain = cq;
            hStatus = B.seed_status[cq]; hRead = B.chain_read[cq]; hContig = B.chain_contig[cq]; hPos = B.chain_pos[cq]; hOff = B.chain_offset[cq];
            hCg0 = B.cigar_off[cq]; hCg1 = B.cigar_off[cq + 1];
        }
        int hR0 = 0, hR1 = 0, hC0lo = 0, hC0hi = 0, hC1lo = 0, hC1hi = 0;
        if(hStatus == HLALA_CHAIN_OK) {
            hR0 = B.read_off[hRead]; hR1 = B.read_off[hRead + 1];
            const long long a0 = contig_off[hContig], a1 = contig_off[hContig + 1];
            hC0lo = (int)(u32)a0; hC0hi = (int)(a0 >> 32); hC1lo = (int)(u32)a1; hC1hi = (int)(a1 >> 32);
        }
        for(int drawn = c0; drawn < cEnd; drawn++) {
        const int hq = drawn - c0;
        const int c = __builtin_amdgcn_readlane(hChain, hq);
        if(__builtin_amdgcn_readlane(hStatus, hq) == HLALA_CHAIN_OK) {
        const int r = __builtin_amdgcn_readlane(hRead, hq);
        const int rOff = __builtin_amdgcn_readlane(hR0, hq), readLen = __builtin_amdgcn_readlane(hR1, hq) - rOff;
        const int contig = __builtin_amdgcn_readlane(hContig, hq);
        const long long cOff = (long long)(((u64)(u32)__builtin_amdgcn_readlane(hC0hi, hq) << 32) | (u64)(u32)__builtin_amdgcn_readlane(hC0lo, hq));
        const long long cLen = (long long)(((u64)(u32)__builtin_amdgcn_readlane(hC1hi, hq) << 32) | (u64)(u32)__builtin_amdgcn_readlane(hC1lo, hq)) - cOff;
        const int pos = __builtin_amdgcn_readlane(hPos, hq), tOffset = __builtin_amdgcn_readlane(hOff, hq);
        const int cg0 = __builtin_amdgcn_readlane(hCg0, hq), nOps = __builtin_amdgcn_readlane(hCg1, hq) - cg0;
        typedef LvCodec<PL> LC;
        // 16-bit layouts: column levels are kept as offsets from the first DEFINED level at or after the chain's first reference position (a translation
        // table holds -1 where a contig base is on no graph level, processBAM.cpp:2519, 5293: the chain may well start on one)
        // (the first 64 positions and the first 64 CIGAR operations are requested together: one round trip, not two)
        int lvBase = 0;
        const u32 cgFirst = lane < nOps ? B.cigar[cg0 + lane] : 0;
        if constexpr (LC::SHORT) {
            const long long t0 = (long long)pos - tOffset;
            for(long long q0 = t0 > 0 ? t0 : 0, qEnd = q0 + 2 * (long long)PL::CAP; q0 < cLen && q0 < qEnd; q0 += 64) {
                const long long q = q0 + lane;
                const int lv = q < cLen ? contig_level[cOff + q] : -1;
                const int m = -wave_max_i32(lv >= 0 ? -lv : -0x7FFFFFFF);
                if(m != 0x7FFFFFFF) { lvBase = m; break; }
            }
        }
        if(lane == 0) { P.err = 0; }
        WSYNC();
        long long tPh[7] = {0, 0, 0, 0, 0, 0, 0};
        long long tFine0[12], tSub0[4]; int dbgN[8] = {0, 0, 0, 0, 0, 0, 0, 0};      // (HLALA_DEBUG) what a heavy read looked like: operations, columns, padded columns, levels, window nodes, segments, long ranges, form
        if constexpr (PL::LONG && PJ_FINE) if(B.dbg) { for(int i = 0; i < 12; i++) tFine0[i] = tFine[i]; for(int i = 0; i < 4; i++) tSub0[i] = tSub[i]; }

        PJ_T(0); PJ_F(-1);
        // ---------------- CIGAR walk (transformBAMreadToInternalAlignment, :4794-5337)
        // Columns are the M/=/X/D/I operations in CIGAR order: M -> (ref, read), D -> (ref, '_'), I -> (-1, '_', read);
        // S advances the read index, H only counts at the very start (:4868-4874), P is dropped (:4814-4828), N throws (:5167).
        if(nOps < 1) { if(lane == 0) PJ_FAIL(HLALA_CHAIN_ERR_INPUT); }
        int nCols = 0;
        // 64 CIGAR operations at a time, one per lane; the per-operation starts stay in registers and are broadcast with readlane.
        // (Short reads: a single round.  Records with more operations carry the running column / reference / read offsets from round to round.)
        if(PJ_OK()) {
            int baseCol = 0, baseRef = 0, baseRead = 0, leadH = 0, prevOp = -1;
            int firstStart = -1, lastRS = 0, lastLen = 0, lastUse = 0;
            // pass 1: totals, validity, first / last column operation (sequence_aligned_{start,stop}InRaw, :5197-5203)
            for(int ob = 0; ob < nOps; ob += 64) {
                const int oi = ob + lane;
                u32 cg = ob == 0 ? cgFirst : (oi < nOps ? B.cigar[cg0 + oi] : 0);
                int op = (int)(cg & 15u), len = (int)(cg >> 4);
                if(oi >= nOps) { op = 6; len = 0; }                       // behaves like 'P'
                bool isCol = (op == 0 || op == 7 || op == 8 || op == 2 || op == 1);
                bool useRef = (op == 0 || op == 7 || op == 8 || op == 2);
                bool useRead = (op == 0 || op == 7 || op == 8 || op == 1 || op == 4);
                if(ob == 0) { int lh = 0; if(lane == 0 && op == 5) lh = len; leadH = __shfl(lh, 0); }     // leading hard clip offsets the unclipped read index
                int tc, tr, tq;
                int colStart = baseCol + wave_excl_scan(isCol ? len : 0, tc);
                int refStart = baseRef + wave_excl_scan(useRef ? len : 0, tr);
                int readStart = baseRead + wave_excl_scan(useRead ? len : 0, tq) + leadH;
                int pop = __shfl_up(op, 1); if(lane == 0) pop = prevOp;          // the operation before this one (previous round for lane 0)
                if(oi < nOps) {
                    if(op == 3 || op > 8) PJ_FAIL(HLALA_CHAIN_ERR_INPUT);
                    // an insertion must follow a column operation or open the alignment (assert(index_along_read == 0), :5086)
                    if(op == 1 && colStart > 0 && oi > 0) {
                        if(!(pop == 0 || pop == 7 || pop == 8 || pop == 2 || pop == 1)) PJ_FAIL(HLALA_CHAIN_ERR_INPUT);
                    }
                }
                prevOp = __shfl(op, 63);
                u64 colMask = __ballot(isCol && len > 0);
                if(colMask) {
                    int firstOp = __ffsll((long long)colMask) - 1, lastOp = 63 - __clzll((long long)colMask);
                    if(firstStart < 0) firstStart = __shfl(readStart, firstOp);
                    lastRS = __shfl(readStart, lastOp); lastLen = __shfl(len, lastOp); lastUse = __shfl(useRead ? 1 : 0, lastOp);
                }
                baseCol += tc; baseRef += tr; baseRead += tq;
            }
            nCols = baseCol;
            if(firstStart < 0) { if(lane == 0) PJ_FAIL(HLALA_CHAIN_ERR_INPUT); }
            else if(lane == 0) { P.startRaw = firstStart; P.stopRaw = lastRS + (lastUse ? lastLen : 0) - 1; }
            if(nCols > PL::CAP || nCols > stride) { if(lane == 0) PJ_FAIL(HLALA_CHAIN_ERR_COLUMNS); }
        }
        WSYNC(); PJ_F(9);
        if(PJ_OK()) {
            // pass 2: columns.  M/=/X/D/I operations in CIGAR order: M -> (ref, read), D -> (ref, '_'), I -> (-1, '_', read)
            int baseCol = 0, baseRef = 0, baseRead = 0, leadH = 0;
            for(int ob = 0; ob < nOps; ob += 64) {
                const int oi = ob + lane;
                u32 cg = ob == 0 ? cgFirst : (oi < nOps ? B.cigar[cg0 + oi] : 0);
                int opK = (int)(cg & 15u), opLen = (int)(cg >> 4);
                if(oi >= nOps) { opK = 6; opLen = 0; }
                bool isCol = (opK == 0 || opK == 7 || opK == 8 || opK == 2 || opK == 1);
                bool useRef = (opK == 0 || opK == 7 || opK == 8 || opK == 2);
                bool useRead = (opK == 0 || opK == 7 || opK == 8 || opK == 1 || opK == 4);
                if(ob == 0) { int lh = 0; if(lane == 0 && opK == 5) lh = opLen; leadH = __shfl(lh, 0); }
                int tc, tr, tq;
                int opCol = baseCol + wave_excl_scan(isCol ? opLen : 0, tc);
                int opRef = baseRef + wave_excl_scan(useRef ? opLen : 0, tr);
                int opRead = baseRead + wave_excl_scan(useRead ? opLen : 0, tq) + leadH;
                // one lane per COLUMN, PJ_U x 64 columns at a time (a 2 x 150 bp chain in one go): the lane finds the operation of each of its columns among
                // the (wave-uniform) operations that overlap them, then all gathers are in flight together.  (One operation after the other cost a round
                // trip of dependent loads per operation -- gene-window alignments carry a dozen --, 64 columns at a time one per 64 columns.)
                constexpr int PJ_U = PL::LONG ? 4 : 6;
                const int nHere = min(64, nOps - ob);
                for(int j0 = 0; j0 < tc; j0 += 64 * PJ_U) {
                    int myOp[PJ_U], refB[PJ_U], readB[PJ_U];          // operation of column j0 + 64 u + lane; reference / read offset of the column = base + column
                    #pragma unroll
                    for(int u = 0; u < PJ_U; u++) { myOp[u] = -1; refB[u] = 0; readB[u] = 0; }
                    if constexpr (PL::LONG) {
                        // Long reads carry thousands of operations: every round of 64 walked all of them once per block of columns, a chain of scalar reads and branches
                        // (1.8 M of a read's 10 M cycles).  The operations' first columns ascend over the lanes, so each lane finds the operation of its column by bisection --
                        // six lane permutes -- and fetches its fields with four more.  (An operation without columns shares its first column with the next one: the LAST
                        // operation that starts at or before a column is the one that holds it.)
                        const int colRel = opCol - baseCol;
                        #pragma unroll
                        for(int u = 0; u < PJ_U; u++) {
                            const int jr = j0 + 64 * u + lane;
                            int lo = 0;
                            #pragma unroll
                            for(int step = 32; step >= 1; step >>= 1) { const int cand = lo + step; const int v = __builtin_amdgcn_ds_bpermute(cand << 2, colRel); if(v <= jr) lo = cand; }
                            const int op = __builtin_amdgcn_ds_bpermute(lo << 2, opK), olen = __builtin_amdgcn_ds_bpermute(lo << 2, opLen), ocs = __builtin_amdgcn_ds_bpermute(lo << 2, colRel);
                            const int orf = __builtin_amdgcn_ds_bpermute(lo << 2, opRef), ord = __builtin_amdgcn_ds_bpermute(lo << 2, opRead);
                            if(jr < tc && (op == 0 || op == 7 || op == 8 || op == 2 || op == 1) && jr >= ocs && jr < ocs + olen) { myOp[u] = op; refB[u] = orf - ocs; readB[u] = ord - ocs; }
                        }
                    } else
                    for(int o = 0; o < nHere; o++) {
                        const int op = __builtin_amdgcn_readlane(opK, o), olen = __builtin_amdgcn_readlane(opLen, o);
                        if(!(op == 0 || op == 7 || op == 8 || op == 2 || op == 1) || olen == 0) continue;
                        const int ocs = __builtin_amdgcn_readlane(opCol, o) - baseCol;
                        if(ocs + olen <= j0 || ocs >= j0 + 64 * PJ_U) continue;
                        const int oref = __builtin_amdgcn_readlane(opRef, o) - ocs, oread = __builtin_amdgcn_readlane(opRead, o) - ocs;
                        #pragma unroll
                        for(int u = 0; u < PJ_U; u++) { const int jr = j0 + 64 * u + lane; if(jr >= ocs && jr < ocs + olen) { myOp[u] = op; refB[u] = oref; readB[u] = oread; } }
                    }
                    int lvv[PJ_U]; unsigned char gcv[PJ_U], scv[PJ_U]; int bad = 0;
                    #pragma unroll
                    for(int u = 0; u < PJ_U; u++) {
                        lvv[u] = -1; gcv[u] = '_'; scv[u] = '_';
                        const int jr = j0 + 64 * u + lane;
                        if(myOp[u] >= 0 && myOp[u] != 1) {
                            const int refpos = pos + refB[u] + jr, ti = refpos - tOffset;
                            if(refpos < 0 || refpos >= cLen || ti < 0 || ti >= cLen) bad = 1;
                            else { gcv[u] = contig_seq[cOff + refpos]; lvv[u] = contig_level[cOff + ti]; }
                        }
                        if(myOp[u] >= 0 && myOp[u] != 2) {
                            const int ri = readB[u] + jr;
                            if(ri < 0 || ri >= readLen) bad = 1; else scv[u] = B.read_bases[rOff + ri];
                        }
                    }
                    if(bad) PJ_FAIL(HLALA_CHAIN_ERR_INPUT);
                    #pragma unroll
                    for(int u = 0; u < PJ_U; u++) {
                        if(myOp[u] >= 0) {
                            int lv = lvv[u];
                            const int j = baseCol + j0 + 64 * u + lane;
                            if constexpr (LC::SHORT) { if(lv >= 0 && lv < lvBase) { PJ_FAIL(HLALA_CHAIN_ERR_INPUT); lv = -1; } else if(lv - lvBase >= 65534) { PJ_FAIL(HLALA_CHAIN_ERR_COLUMNS); lv = -1; } }
                            P.lvl[0][j] = LC::put(lv, lvBase); P.g[0][j] = gcv[u]; P.s[0][j] = scv[u];
                        }
                    }
                }
                baseCol += tc; baseRef += tr; baseRead += tq;
            }
        }
        WSYNC();
        if(PJ_OK() && !(uni(P.startRaw) < uni(P.stopRaw))) { if(lane == 0) PJ_FAIL(HLALA_CHAIN_ERR_INPUT); }        // :5252
        WSYNC();

        if constexpr (PL::LONG && PJ_FINE) if(B.dbg) { dbgN[0] = nOps; dbgN[1] = nCols; }
        PJ_F(10);
        PJ_T(1);
        // ---------------- trim leading / trailing insertion columns, pad skipped levels (:2518-2579)
        int n1 = 0;
        if(PJ_OK()) {
            int firstCol = nCols, lastCol = -1;
            for(int j0 = 0; j0 < nCols; j0 += 64) {
                int j = j0 + lane; bool def = j < nCols && LC::get(P.lvl[0][j], lvBase) != -1;
                u64 m = __ballot(def);
                if(m) { int f = j0 + __ffsll((long long)m) - 1, l = j0 + 63 - __clzll((long long)m); if(f < firstCol) firstCol = f; if(l > lastCol) lastCol = l; }
            }
            if(lastCol < 0 || !(firstCol < lastCol)) { if(lane == 0) PJ_FAIL(HLALA_CHAIN_ERR_INPUT); }       // "all insertions" :5271 / assert :2535
            else {
                if(lane == 0) { P.startRaw += firstCol; P.stopRaw -= (nCols - 1 - lastCol); }
                // prefix over columns: new index = (j - firstCol) + sum of level gaps up to and including j
                int carryGap = 0, carryPrev = -1;
                for(int j0 = firstCol; j0 <= lastCol; j0 += 64) {
                    int j = j0 + lane; bool act = j <= lastCol;
                    int lv = act ? LC::get(P.lvl[0][j], lvBase) : -1;
                    // last defined level strictly before j: prefix max (levels increase along the alignment)
                    int pm = lv;
                    for(int o = 1; o < 64; o <<= 1) { int y = __shfl_up(pm, o); if(lane >= o) pm = max(pm, y); }
                    int prevIncl = max(pm, carryPrev);
                    int prevExcl = __shfl_up(prevIncl, 1); if(lane == 0) prevExcl = carryPrev;
                    int gap = 0;
                    if(act && lv != -1 && prevExcl != -1) { gap = lv - prevExcl - 1; if(gap < 0) { PJ_FAIL(HLALA_CHAIN_ERR_INPUT); gap = 0; } }
                    int tg; int gb = wave_excl_scan(gap, tg);
                    int np = (j - firstCol) + carryGap + gb + gap;
                    if(act) {
                        if(np >= PL::CAP || np >= stride) PJ_FAIL(HLALA_CHAIN_ERR_COLUMNS);
                        else {
                            for(int q = 0; q < gap; q++) { int w = np - gap + q; P.lvl[1][w] = LC::put(prevExcl + 1 + q, lvBase); P.g[1][w] = '_'; P.s[1][w] = '_'; }
                            P.lvl[1][np] = LC::put(lv, lvBase); P.g[1][np] = P.g[0][j]; P.s[1][np] = P.s[0][j];
                        }
                    }
                    carryGap += tg; carryPrev = __shfl(prevIncl, 63);
                }
                n1 = (lastCol - firstCol + 1) + carryGap;
                if(n1 > PL::CAP || n1 > stride) { if(lane == 0) PJ_FAIL(HLALA_CHAIN_ERR_COLUMNS); }
            }
        }
        WSYNC();
        int cur = 1;       // buffer holding the current columns

        PJ_T(2);
        // ---------------- cleanInitialAlignment (:4621-4792) -- only when an insertion or a double-gap column exists
        // Bit-mask helpers over 64-column words in LDS; every lane runs the same scalar walk (uniform control flow).
        auto bitOf = [&](const u64* m, int p) -> bool { return (m[p >> 6] >> (p & 63)) & 1ull; };
        auto nextSet = [&](const u64* m, int p, int hi) -> int {                          // first set bit in [p, hi], else hi + 1
            while(p <= hi) { u64 w = m[p >> 6] >> (p & 63); if(w) { int q = p + __ffsll((long long)w) - 1; return q <= hi ? q : hi + 1; } p = (p | 63) + 1; }
            return hi + 1; };
        auto nextClear = [&](const u64* m, int p, int hi) -> int {                        // first clear bit in [p, hi], else hi + 1
            while(p <= hi) { u64 w = (~m[p >> 6]) >> (p & 63); if(w) { int q = p + __ffsll((long long)w) - 1; return q <= hi ? q : hi + 1; } p = (p | 63) + 1; }
            return hi + 1; };
        auto prevSet = [&](const u64* m, int p, int lo) -> int {                          // last set bit in [lo, p], else lo - 1
            while(p >= lo) { u64 w = m[p >> 6] << (63 - (p & 63)); if(w) { int q = p - __clzll((long long)w); return q >= lo ? q : lo - 1; } p = (p & ~63) - 1; }
            return lo - 1; };
        auto countBits = [&](const u64* m, int a, int b) -> int {                         // set bits in [a, b]
            int n = 0;
            for(int k = (a >> 6); k <= (b >> 6) && a <= b; k++) {
                u64 w = m[k]; int lo = k * 64, hi = lo + 63;
                if(a > lo) w &= ~0ull << (a - lo);
                if(b < hi) w &= ~0ull >> (hi - b);
                n += __popcll(w);
            }
            return n; };
        int removed = 0;
        if(PJ_OK()) {
            // masks: mGap = interesting column (insertion or double gap), mDef = insertion (level -1), mSeq = double gap
            bool any = false;
            const int nW = (n1 + 63) >> 6;
            for(int k = 0; k < nW; k++) {
                int j = k * 64 + lane; bool ins = false, dg = false;
                if(j < n1) { ins = LC::get(P.lvl[cur][j], lvBase) == -1; dg = P.g[cur][j] == '_' && P.s[cur][j] == '_'; }
                u64 mi = __ballot(ins), md = __ballot(dg);
                if(mi | md) any = true;
                if(lane == 0) { P.mGap[k] = mi | md; P.mDef[k] = mi; P.mSeq[k] = md; }
            }
            WSYNC();
            if(any) {
                bool cleaned = false;
                int p = 0;
                while(p < n1) {
                    const int a = nextSet(P.mGap, p, n1 - 1);
                    if(a > n1 - 1) break;
                    const int e = nextClear(P.mGap, a, n1 - 1);
                    if(e > n1 - 1) break;                                   // a stretch that reaches the last column is never closed (:4650-4720)
                    const int b = e - 1;
                    const int ni = countBits(P.mDef, a, b), nd = countBits(P.mSeq, a, b);
                    if(ni - nd == 0) {
                        // pair the inserted bases with the skipped levels: first half = (gap level, '_', base), rest deleted
                        const int Ls = b - a + 1, half = Ls / 2, ng = Ls - ni;
                        if(ni != ng || ni != half) { if(lane == 0) PJ_FAIL(HLALA_CHAIN_ERR_INPUT); }
                        else {
                            cleaned = true;
                            auto olv = P.lvl[1 - cur]; auto osa = P.s[1 - cur];      // scratch: gathered in column order
                            for(int q = a + lane; q <= b; q += 64) {
                                const int r = q > a ? countBits(P.mDef, a, q - 1) : 0;
                                if(bitOf(P.mDef, q)) osa[r] = P.s[cur][q]; else olv[(q - a) - r] = P.lvl[cur][q];
                            }
                            WSYNC();
                            for(int q = a + lane; q <= b; q += 64) {
                                const int i = q - a;
                                if(i < half) { P.lvl[cur][q] = olv[i]; P.g[cur][q] = '_'; P.s[cur][q] = osa[i]; }
                                else { P.lvl[cur][q] = LC::put(-1, lvBase); P.g[cur][q] = '_'; P.s[cur][q] = '_'; }
                            }
                            WSYNC();
                        }
                    }
                    p = e;
                }
                if(cleaned) {
                    int w = 0;
                    for(int k = 0; k < nW; k++) {
                        int j = k * 64 + lane; bool keep = false; int lv = 0; unsigned char gc = 0, sc = 0;
                        if(j < n1) { lv = LC::get(P.lvl[cur][j], lvBase); gc = P.g[cur][j]; sc = P.s[cur][j]; keep = !(lv == -1 && gc == '_' && sc == '_'); }
                        const u64 m = __ballot(keep);
                        if(keep) { const int o = w + __popcll(m & ((1ull << lane) - 1ull)); P.lvl[1 - cur][o] = LC::put(lv, lvBase); P.g[1 - cur][o] = gc; P.s[1 - cur][o] = sc; }
                        w += __popcll(m);
                    }
                    WSYNC();
                    cur = 1 - cur; n1 = w;
                }
            }
        }
        WSYNC();

        PJ_T(3);
        // ---------------- restrictInitialAlignmentToNoGapAreas (:4461-4619) -- only when a gap-stretch level is touched
        if(PJ_OK()) {
            // three bit masks per 64 columns (gap-stretch level / defined level / read character) carry all the step needs
            bool any = false; int seqChars = 0;
            const int nW = (n1 + 63) >> 6;
            // (the gap-stretch flags of eight words of columns are requested together: the ballots made each word a round trip of its own)
            constexpr int PJ_W = PL::LONG ? 4 : 8;
            const int nLevelsG = G.L;
            for(int k0 = 0; k0 < nW; k0 += PJ_W) {
                unsigned char gs[PJ_W]; bool defv[PJ_W], sqv[PJ_W]; int bad = 0;
                #pragma unroll
                for(int u = 0; u < PJ_W; u++) {
                    const int j = (k0 + u) * 64 + lane; gs[u] = 0; defv[u] = false; sqv[u] = false;
                    if(k0 + u < nW && j < n1) {
                        const int l = LC::get(P.lvl[cur][j], lvBase); defv[u] = (l != -1); sqv[u] = (P.s[cur][j] != '_');
                        if(defv[u]) { if(l < 0 || l >= nLevelsG - 1) bad = 1; else gs[u] = G.gap_stretch[l]; }
                    }
                }
                if(bad) PJ_FAIL(HLALA_CHAIN_ERR_INPUT);
                #pragma unroll
                for(int u = 0; u < PJ_W; u++) {
                    if(k0 + u < nW) {
                        const u64 mg = __ballot(gs[u] != 0), md = __ballot(defv[u]), ms = __ballot(sqv[u]);
                        if(mg) any = true;
                        seqChars += __popcll(ms);
                        if(lane == 0) { P.mGap[k0 + u] = mg; P.mDef[k0 + u] = md; P.mSeq[k0 + u] = ms; }
                    }
                }
            }
            WSYNC();
            if(any && PJ_OK()) {
                // every lane runs the same scalar walk over the masks (uniform control flow, no divergence)
                int bestA = -1, bestB = -1, bestLen = 0;
                auto consider = [&](int a, int b) {                                         // trim -1 ends (:4507-4531), keep the LAST longest (:4533-4552)
                    int a2 = nextSet(P.mDef, a, b);
                    if(a2 <= b) { int b2 = prevSet(P.mDef, b, a2); int len = b2 - a2 + 1; if(len >= bestLen) { bestLen = len; bestA = a2; bestB = b2; } }
                };
                // maximal stretches without a gap-stretch level; a gap column ends the stretch before it (:4470-4502)
                int p = 0;
                while(p < n1) {
                    if(bitOf(P.mGap, p)) { p++; continue; }
                    int q = nextSet(P.mGap, p, n1 - 1);                                      // stretch = [p, q - 1]
                    if(!(p == 0 && q == n1)) consider(p, q - 1);                             // a stretch from column 0 to the end is not recorded (:4492-4502)
                    p = q;
                }
                int w = n1;
                if(bestLen > 0) {
                    int ns = uni(P.startRaw) + countBits(P.mSeq, 0, bestA - 1), ne = uni(P.stopRaw) - countBits(P.mSeq, bestB + 1, n1 - 1), sc = countBits(P.mSeq, bestA, bestB);
                    if(((double)sc / (double)seqChars) > 0.3) {                              // :4606
                        for(int q = bestA + lane; q <= bestB; q += 64) { P.lvl[1 - cur][q - bestA] = P.lvl[cur][q]; P.g[1 - cur][q - bestA] = P.g[cur][q]; P.s[1 - cur][q - bestA] = P.s[cur][q]; }
                        WSYNC();
                        cur = 1 - cur; w = bestLen;
                        if(lane == 0) { P.startRaw = ns; P.stopRaw = ne; }
                    }
                }
                removed = n1 - w; n1 = w;
            }
        }
        WSYNC();

        PJ_T(4); PJ_F(-1);
        // ---------------- re-threading DP, sequence variant (:2676-2835)
        // state of column i = best number of edge labels equal to the read character over all graph paths that respect
        // the seed's matches, per node of the column's target level; ties keep the smallest edge (std::set<Edge*> order).
        //
        // Device formulation.  The chain's window of the in-edge CSR is staged into LDS.  A level that holds a single node is a
        // cut: every path passes through it, so the recurrence to the right of it only sees the left part as an additive
        // constant, and neither the argmax per node nor the order of the ties depends on that constant.  The window is
        // therefore split at its single-node levels into independent segments (S = 0 at each segment start) that are solved
        // one per lane, and the backtrace likewise restarts at every cut node.  In a PRG most levels are single-node, so
        // segments are a few levels long.  Windows that do not fit the LDS staging, that contain a segment longer than
        // PROJ_SEGMAX levels, or in which a segment hits the "no node reachable" assert, run the column-sequential form below.
        int level0 = -1, nb = 0, chCount = 0; ChoiceRec* ch = slabCh;
        int nDef = 0, nodeBase = 0, eBase = 0, nEdges = 0; bool staged = false, par = false, windowed = false, chunked = false, deferred = false; int defCount = 0, nChunks = 0;
        int nLong = 0; bool lastLong = false;      // (long reads) level ranges solved level by level; the last of them ends the window
        if(PJ_OK()) {
            level0 = uni(LC::get(P.lvl[cur][0], lvBase));
            int lastLevel = uni(LC::get(P.lvl[cur][n1 - 1], lvBase));
            if(level0 < 0 || lastLevel < level0 || lastLevel + 1 >= G.L) { if(lane == 0) PJ_FAIL(HLALA_CHAIN_ERR_INPUT); }
            else {
                nDef = lastLevel - level0 + 1;
                int v = 0;                                                                     // four offsets, one round trip
                if(lane == 0) v = G.level_off[level0]; else if(lane == 1) v = G.level_off[level0 + 1]; else if(lane == 2) v = G.level_off[lastLevel + 2];
                nodeBase = __builtin_amdgcn_readlane(v, 0); nb = __builtin_amdgcn_readlane(v, 1);
                const int nodeEnd = __builtin_amdgcn_readlane(v, 2);
                chCount = nodeEnd - nb;
                int v2 = 0; if(lane == 0) v2 = G.in_off[nb]; else if(lane == 1) v2 = G.in_off[nodeEnd];
                eBase = __builtin_amdgcn_readlane(v2, 0);
                nEdges = __builtin_amdgcn_readlane(v2, 1) - eBase;
                staged = (nDef <= PL::CAP) && (nodeEnd - nodeBase <= PL::SN) && (nEdges <= PL::SE);
                // offsets of the window's levels (16-bit, relative to its first node): the segment-parallel and the chunked form both need them
                windowed = (nDef <= PL::CAP) && (nodeEnd - nodeBase < 65536) && (nEdges < 65536);
                if constexpr (!PL::LONG) {
                    // level offsets and -- where the window fits the staging arrays -- its in-edge CSR: every load is requested before the first store (one round trip)
                    constexpr int U1 = (PL::CAP + 2 + 63) / 64, U2 = (PL::SN + 1 + 63) / 64, U3 = (PL::SE + 63) / 64;
                    int v1[U1], v2[U2]; FromLab v3[U3];
                    if(windowed) {
                        #pragma unroll
                        for(int u = 0; u < U1; u++) { const int i = lane + 64 * u; if(i <= nDef + 1) v1[u] = G.level_off[level0 + i]; }
                    }
                    if(staged) {
                        #pragma unroll
                        for(int u = 0; u < U2; u++) { const int i = lane + 64 * u; if(i <= chCount) v2[u] = G.in_off[nb + i]; }
                        #pragma unroll
                        for(int u = 0; u < U3; u++) { const int e = lane + 64 * u; if(e < nEdges) { v3[u].from = G.in_from[eBase + e]; v3[u].lab = G.in_label[eBase + e]; } }
                    }
                    if(windowed) {
                        #pragma unroll
                        for(int u = 0; u < U1; u++) { const int i = lane + 64 * u; if(i <= nDef + 1) P.sLev[i] = (unsigned short)(v1[u] - nodeBase); }
                    }
                    if(staged) {
                        #pragma unroll
                        for(int u = 0; u < U2; u++) { const int i = lane + 64 * u; if(i <= chCount) P.sIn[i] = (unsigned short)(v2[u] - eBase); }
                        #pragma unroll
                        for(int u = 0; u < U3; u++) { const int e = lane + 64 * u; if(e < nEdges) { P.sFrom[e] = (unsigned short)(v3[u].from - nodeBase); P.sLab[e] = v3[u].lab; } }
                    }
                } else {
                if(windowed) staged_rows<PJL_U>(lane, nDef + 2, 64, [&](int i) { return G.level_off[level0 + i]; }, [&](int i, int v) { P.sLev[i] = (unsigned short)(v - nodeBase); });
                if(staged) {
                    staged_rows<PJL_U>(lane, chCount + 1, 64, [&](int i) { return G.in_off[nb + i]; }, [&](int i, int v) { P.sIn[i] = (unsigned short)(v - eBase); });
                    staged_rows<PJL_U>(lane, nEdges, 64, [&](int e) { FromLab r; r.from = G.in_from[eBase + e]; r.lab = G.in_label[eBase + e]; return r; },
                                   [&](int e, FromLab r) { P.sFrom[e] = (unsigned short)(r.from - nodeBase); P.sLab[e] = r.lab; });
                }
                }
            }
        }
        WSYNC(); PJ_F(0);
        u64 edgesTouched = 0;
        int nSeg = 0;
        auto const Sflat = P.sflat();                                                    // S per node of the window (parallel form)
        const bool haveCols = windowed || (PL::LONG && nDef <= PL::CAP);      // (long reads: the level-by-level form needs the table whatever the size of the window)
        if(PJ_OK() && haveCols) {
            // level -> (column, read character, seed-is-match); the defined columns must cover level0..lastLevel exactly once
            for(int j0 = 0; j0 < n1; j0 += 64) {
                int j = j0 + lane; bool d = false;
                if(j < n1) {
                    int l = LC::get(P.lvl[cur][j], lvBase);
                    if(l != -1) {
                        d = true; int li = l - level0;
                        if(li >= 0 && li < nDef) { u32 sc = P.s[cur][j], gc = P.g[cur][j]; P.colInfo[li] = (u32)j | (sc << 16) | ((sc == gc ? 1u : 0u) << 24); }
                    }
                }
                defCount += __popcll(__ballot(d));
            }
        }
        WSYNC(); PJ_F(1);
        if(PJ_OK() && staged) {
            for(int i0 = 0; i0 < nDef; i0 += 64) {
                int i = i0 + lane;
                bool cut = i < nDef && (i == 0 || (P.sLev[i + 1] - P.sLev[i]) == 1);
                u64 m = __ballot(cut);
                if(cut) P.segStart[nSeg + __popcll(m & ((1ull << lane) - 1ull))] = (unsigned short)i;
                nSeg += __popcll(m);
            }
            if(lane == 0) P.segStart[nSeg] = (unsigned short)nDef;
            WSYNC();
            int mx = 0;
            for(int sg = lane; sg < nSeg; sg += 64) mx = max(mx, (int)P.segStart[sg + 1] - (int)P.segStart[sg]);
            mx = wave_max_i32(mx);
            // Long reads (round 6): a read that crosses a gene window has a segment of thousands of levels between its thousands of short ones.  The short ones are solved one per lane
            // as in every other read, the long ones (listed here) level by level further down -- the whole read only when the list overflows or its window is not staged.
            if constexpr (PL::LONG) if(mx > PROJ_SEGMAX) {
                for(int sg0 = 0; sg0 < nSeg; sg0 += 64) {
                    const int sgl = sg0 + lane;
                    const int sa = sgl < nSeg ? (int)P.segStart[sgl] : 0, sb = sgl < nSeg ? (int)P.segStart[sgl + 1] : 0;
                    const bool isLong = sb - sa > PROJ_SEGMAX;
                    const u64 lm = __ballot(isLong);
                    const int pos = nLong + (int)__popcll(lm & ((1ull << lane) - 1ull));
                    if(isLong && pos < PROJL_LONGSEG) { P.longSeg[2 * pos] = sa; P.longSeg[2 * pos + 1] = sb; }
                    nLong += (int)__popcll(lm);
                }
            }
            par = (defCount == nDef) && (mx <= PROJ_SEGMAX || (PL::LONG && nLong <= min(PROJL_LONGSEG, B.long_max_segs)));
            PJ_F(2);
            if(par) {
                const int nbR = nb - nodeBase;
                int fail = 0;
                for(int sg = lane; sg < nSeg; sg += 64) {
                    const int a = P.segStart[sg], b = P.segStart[sg + 1];
                    if(PL::LONG && b - a > PROJ_SEGMAX) continue;
                    for(int i = a; i < b && !fail; i++) {
                        const u32 ci = P.colInfo[i];
                        const unsigned char sc = (unsigned char)((ci >> 16) & 0xFFu); const bool seedIsMatch = ((ci >> 24) & 1u) != 0;
                        const int t0 = P.sLev[i + 1], t1 = P.sLev[i + 2];
                        int anyReached = 0;
                        for(int t = t0; t < t1; t++) {
                            int best = -1, bestE = 0xFFFF;
                            const int e0 = P.sIn[t - nbR], e1 = P.sIn[t - nbR + 1];
                            for(int e = e0; e < e1; e++) {                                   // in-edges in creation order: first maximum = smallest edge
                                int sp = (i == a) ? 0 : (int)Sflat[P.sFrom[e]];
                                if(sp < 0) continue;
                                unsigned char lab = P.sLab[e];
                                if(seedIsMatch && lab != sc) continue;                        // :2803-2809
                                int cand = sp + (lab == sc ? 1 : 0);
                                if(cand > best) { best = cand; bestE = e; }
                            }
                            Sflat[t] = (short)best;
                            P.sChoice[t - nbR] = (unsigned short)bestE;
                            if(best >= 0) anyReached = 1;
                        }
                        if(!anyReached) fail = 1;
                    }
                }
                PJ_F(3);
                PJ_F(4);
                if(__ballot(fail)) par = false;                                               // let the sequential form raise the reference's assert
                else if(lane == 0) edgesTouched += (u64)nEdges;
            }
        }
        WSYNC();
        if(PJ_OK() && !par) {
            if(chCount > slabEnt) { if(lane == 0) PJ_FAIL(HLALA_CHAIN_ERR_FRONTIER); }
            int m0 = nb - nodeBase;
            if(m0 > PROJ_NODES) { if(lane == 0) PJ_FAIL(HLALA_CHAIN_ERR_FRONTIER); }
            else for(int z = lane; z < m0; z += 64) P.Srow[0][z] = 0;                         // all nodes of the first level, S = 0 (:2694-2701)
        }
        WSYNC();
        if(PJ_OK()) {
            int rowP = 0;
            bool fastLong = false;       // (long reads) the whole read level by level
            bool runLevels = false;
            if constexpr (PL::LONG) {
                if(!par) nLong = 0;
                if(par) runLevels = nLong > 0;
                else if(haveCols && defCount == nDef) { runLevels = true; fastLong = true; nLong = 1; if(lane == 0) { P.longSeg[0] = 0; P.longSeg[1] = nDef; } }
            }
            if constexpr (PL::LONG) if(runLevels) {
                // Long reads, level-by-level form (round 6).  A read that crosses a gene window has no cut for thousands of levels; it used to walk ALL its 10 000 levels one by one
                // against HBM -- two dependent round trips per level, 150 M cycles for a read that spans a whole window: the LAST wavefront of every batch (12 500 reads took
                // 135 ms, 50 000 took 160).  Now such levels go through the machinery of k_rethread_chains: chunks of up to 62 consecutive levels whose target nodes and
                // in-edges fit the LDS staging arrays (one round trip per chunk, straight from the graph's arrays: absolute offsets, no window tables), the lanes on a level's
                // IN-EDGES (DevGraph::in_rec), scores handed from lane to lane, one 32-bit back pointer per node (from-node rank | in-edge within its level) in the wave's slab.
                u32* const chw = (u32*)slabCh;
                const int chunkNodes = B.long_chunk_nodes > 0 ? min(RT_SN, B.long_chunk_nodes) : RT_SN;
                if(chCount > 2 * slabEnt) { if(lane == 0) PJ_FAIL(HLALA_CHAIN_ERR_FRONTIER); }
                for(int i = lane; i < PROJL_CAP / 64; i += 64) P.chunkStart[i] = 0;
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (the bits are set with atomics at the L2: the zeroes must be there first)
                WSYNC();
                for(int rq = 0; rq < nLong && PJ_OK(); rq++) {
                const int ra = uni(P.longSeg[2 * rq]), rb = uni(P.longSeg[2 * rq + 1]);
                {   // S = 0 on the nodes of the range's first level (:2694-2701; a cut has one)
                    int v = 0; if(lane < 2) v = G.level_off[level0 + ra + lane];
                    const int fromCnt = __builtin_amdgcn_readlane(v, 1) - __builtin_amdgcn_readlane(v, 0);
                    if(fromCnt > PROJ_NODES) { if(lane == 0) PJ_FAIL(HLALA_CHAIN_ERR_FRONTIER); }
                    else for(int z = lane; z < fromCnt; z += 64) P.Srow[rowP][z] = 0;
                    WSYNC();
                }
                lastLong = rb == nDef;
                int a = ra;
                while(a < rb && PJ_OK()) {
                    const long long tC0 = (PJ_FINE && B.dbg) ? clock64() : 0;
                    const int i0 = a + lane;
                    // offsets of levels a .. a + 63 (+ 2), colInfo + level_fast: one round trip; the first in-edges of the levels' target nodes: a second one
                    const int lvA = i0 <= nDef + 1 ? G.level_off[level0 + i0] : 0;
                    const int lvB = i0 + 1 <= nDef + 1 ? G.level_off[level0 + i0 + 1] : 0;
                    const int lvC = i0 + 2 <= nDef + 1 ? G.level_off[level0 + i0 + 2] : 0;
                    const u32 ciReg = i0 < nDef ? ((u32)P.colInfo[i0] | ((u32)G.level_fast[level0 + i0 + 1] << 25)) : 0u;
                    const int sgA = i0 <= nDef ? G.in_off[lvB] : 0;
                    const int sgB = __shfl_down(sgA, 1);
                    const int lvA1 = __builtin_amdgcn_readlane(lvA, 1), sgA0 = __builtin_amdgcn_readlane(sgA, 0);
                    const bool fits = i0 < rb && lane < 62 && (lvC - lvA1) <= chunkNodes && (sgB - sgA0) <= RT_CE;
                    const u64 fm = __ballot(fits);
                    const int cnt = __ffsll((long long)~fm) - 1;                               // levels a .. a + cnt - 1 fit together (a prefix: both sums grow)
                    if(lane == 0) atomicOr(P.chunkStart.p + (a >> 6), 1ull << (a & 63));
                    if(cnt == 0) {
                        // one level wider than the staging arrays: node by node, straight from HBM
                        const u32 ci = (u32)__builtin_amdgcn_readlane((int)ciReg, 0);
                        const unsigned char sc = (unsigned char)((ci >> 16) & 0xFFu); const bool seedIsMatch = ((ci >> 24) & 1u) != 0;
                        const int fb = __builtin_amdgcn_readlane(lvA, 0), tb = lvA1, tm = __builtin_amdgcn_readlane(lvA, 2) - lvA1;
                        if(tm > PROJ_NODES) { if(lane == 0) PJ_FAIL(HLALA_CHAIN_ERR_FRONTIER); break; }
                        int anyReached = 0;
                        for(int z = lane; z < tm; z += 64) {
                            const int node = tb + z;
                            int best = -1, bestE = 0, bestFrom = -1;
                            const int e0 = G.in_off[node], e1 = G.in_off[node + 1];
                            for(int e = e0; e < e1; e++) {
                                const int fz = G.in_from[e] - fb; const int sp = P.Srow[rowP][fz];
                                if(sp < 0) continue;
                                const unsigned char lab = G.in_label[e];
                                if(seedIsMatch && lab != sc) continue;
                                const int cd = sp + (lab == sc ? 1 : 0);
                                if(cd > best) { best = cd; bestE = e - sgA0; bestFrom = fz; }
                            }
                            if(!par) edgesTouched += (u64)(e1 - e0);
                            if(bestE > 0xFFFF) PJ_FAIL(HLALA_CHAIN_ERR_FRONTIER);
                            P.Srow[1 - rowP][z] = (short)best;
                            chw[node - nb] = best >= 0 ? ((u32)bestFrom | ((u32)bestE << 16)) : 0xFFFFFFFFu;
                            if(best >= 0) anyReached = 1;
                        }
                        if(!__ballot(anyReached)) { if(lane == 0) PJ_FAIL(HLALA_CHAIN_ERR_INPUT); break; }
                        rowP = 1 - rowP;
                        WSYNC();
                        a++;
                        continue;
                    }
                    const int b = a + cnt - 1;
                    const int tBase = lvA1, nT = __builtin_amdgcn_readlane(lvA, cnt + 1) - tBase, eC = sgA0, nE = __builtin_amdgcn_readlane(sgA, cnt) - eC;
                    const int lvReg = lvA, sgReg = sgA - eC;
                    {   // the chunk's in-edge records and in-edge offsets: one round trip
                        // (the in-edge offsets of the nodes are only read by the levels that are solved node by node -- mode 0: most chunks have none)
                        const bool needOff = __ballot(lane < cnt && ((ciReg >> 25) & 3u) == 0u) != 0;
                        // (four rows of lanes per round trip: a chunk of backbone levels has one, a chunk in a gene window up to eight)
                        for(int e0 = 0; e0 < nE; e0 += 4 * 64) {
                            u32 ve[4];
                            #pragma unroll
                            for(int u = 0; u < 4; u++) { const int e = e0 + lane + 64 * u; if(e < nE) ve[u] = G.in_rec[eC + e]; }
                            #pragma unroll
                            for(int u = 0; u < 4; u++) { const int e = e0 + lane + 64 * u; if(e < nE) P.cRec[e] = ve[u]; }
                        }
                        if(needOff) for(int t0 = 0; t0 <= nT; t0 += 4 * 64) {
                            int vn[4];
                            #pragma unroll
                            for(int u = 0; u < 4; u++) { const int t = t0 + lane + 64 * u; if(t <= nT) vn[u] = G.in_off[tBase + t]; }
                            #pragma unroll
                            for(int u = 0; u < 4; u++) { const int t = t0 + lane + 64 * u; if(t <= nT) P.cIn[t] = (unsigned short)(vn[u] - eC); }
                        }
                    }
                    WSYNC();
                    const long long tC1 = (PJ_FINE && B.dbg) ? clock64() : 0;
                    bool prevFast = false, stop = false; int sReg = -1;
                    u32 recN = 0;
                    { const int nE0 = __builtin_amdgcn_readlane(sgReg, 1); if(lane < nE0 && nE0 <= 64) recN = P.cRec[lane]; }
                    for(int k = 0; k < cnt; k++) {
                        const u32 ci = (u32)__builtin_amdgcn_readlane((int)ciReg, k);
                        const int sc = (int)((ci >> 16) & 0xFFu); const bool seedIsMatch = ((ci >> 24) & 1u) != 0;
                        const int l1 = __builtin_amdgcn_readlane(lvReg, k + 1), tm = __builtin_amdgcn_readlane(lvReg, k + 2) - l1;
                        const int eL0 = __builtin_amdgcn_readlane(sgReg, k), eL1 = __builtin_amdgcn_readlane(sgReg, k + 1), nEl = eL1 - eL0;
                        const int mode = (int)((ci >> 25) & 3u), maxd = (int)((ci >> 27) & 7u);
                        const u32 rec = recN;
                        if(k + 1 < cnt) { const int nEn = __builtin_amdgcn_readlane(sgReg, k + 2) - eL1; recN = (lane < nEn && nEn <= 64) ? P.cRec[eL1 + lane] : 0u; }
                        bool reached = false;
                        auto edge_key = [&](const u32 r, const int sp, const bool mine) -> int {
                            const int m = (int)((r >> 18) & 0xFFu) == sc ? 1 : 0;
                            const int cand = (mine && sp >= 0 && (m || !seedIsMatch)) ? sp + m + 1 : 0;           // 0: not admitted (:2803-2809) or from-node unreachable; else score + 1
                            return (cand << 15) | ((63 - lane) << 9) | (int)(r & 511u);
                        };
                        auto node_key = [&](const u32 r, const int key) -> int {
                            const int pos = (int)((r >> 15) & 7u);
                            int bk = key, kj = key;
                            for(int j = 1; j <= maxd; j++) { kj = __builtin_amdgcn_update_dpp(0, kj, 0x138, 0xF, 0xF, false); if(pos >= j) bk = max(bk, kj); }          // wave_shr:1 -- the edge j places before
                            return bk;
                        };
                        if(mode == 1) {
                            const bool mine = lane < nEl;
                            int sp;
                            if(prevFast) sp = __builtin_amdgcn_ds_bpermute((int)(((rec >> 9) & 63u) << 2), sReg);
                            else sp = (int)P.Srow[rowP][rec & 511u];
                            const int bk = node_key(rec, edge_key(rec, sp, mine));
                            const bool last = mine && (rec & (1u << 28)) != 0;
                            const u64 lastMask = __ballot(last);
                            const int best = (bk >> 15) - 1;
                            sReg = best;
                            if(last) {
                                const int tz = (int)__builtin_amdgcn_mbcnt_hi((u32)(lastMask >> 32), __builtin_amdgcn_mbcnt_lo((u32)lastMask, 0u));
                                P.Srow[1 - rowP][tz] = (short)best;
                                chw[l1 + tz - nb] = best >= 0 ? ((u32)(bk & 511) | ((u32)(63 - ((bk >> 9) & 63)) << 16)) : 0xFFFFFFFFu;
                                reached = best >= 0;
                            }
                            if(lane == 0 && !par) edgesTouched += (u64)nEl;
                            prevFast = true;
                        } else if(mode == 2) {
                            int nodesDone = 0;
                            for(int s0 = 0; s0 == 0 || s0 + 7 < nEl; s0 += 57) {
                                const bool mine = s0 + lane < nEl;
                                const u32 r = mine ? P.cRec[eL0 + s0 + lane] : 0u;
                                const int sp = (int)P.Srow[rowP][r & 511u];
                                const int bk = node_key(r, edge_key(r, sp, mine));
                                const bool last = mine && (r & (1u << 28)) != 0 && (s0 == 0 || lane >= 7);
                                const u64 lastMask = __ballot(last);
                                if(last) {
                                    const int best = (bk >> 15) - 1;
                                    const int tz = nodesDone + (int)__builtin_amdgcn_mbcnt_hi((u32)(lastMask >> 32), __builtin_amdgcn_mbcnt_lo((u32)lastMask, 0u));
                                    P.Srow[1 - rowP][tz] = (short)best;
                                    chw[l1 + tz - nb] = best >= 0 ? ((u32)(bk & 511) | ((u32)(s0 + 63 - ((bk >> 9) & 63)) << 16)) : 0xFFFFFFFFu;
                                    if(best >= 0) reached = true;
                                }
                                nodesDone += (int)__popcll(lastMask);
                            }
                            if(lane == 0 && !par) edgesTouched += (u64)nEl;
                            prevFast = false;
                        } else {
                            const int t0 = l1 - tBase;
                            for(int z = lane; z < tm; z += 64) {
                                const int t = t0 + z;
                                int best = -1, bestE = 0, bestFrom = -1;
                                const int e0 = P.cIn[t], e1 = P.cIn[t + 1];
                                for(int e = e0; e < e1; e++) {                                   // in-edges in creation order: first maximum = smallest edge
                                    const u32 r = P.cRec[e];
                                    const int fz = (int)(r & 511u);
                                    const int sp = P.Srow[rowP][fz];
                                    if(sp < 0) continue;
                                    const int lab = (int)((r >> 18) & 0xFFu);
                                    if(seedIsMatch && lab != sc) continue;                        // :2803-2809
                                    const int cd = sp + (lab == sc ? 1 : 0);
                                    if(cd > best) { best = cd; bestE = e - eL0; bestFrom = fz; }
                                }
                                if(!par) edgesTouched += (u64)(e1 - e0);
                                P.Srow[1 - rowP][z] = (short)best;
                                chw[tBase + t - nb] = best >= 0 ? ((u32)bestFrom | ((u32)bestE << 16)) : 0xFFFFFFFFu;
                                if(best >= 0) reached = true;
                            }
                            prevFast = false;
                        }
                        if(!__ballot(reached)) { if(lane == 0) PJ_FAIL(HLALA_CHAIN_ERR_INPUT); stop = true; break; }      // assert(seedChain_backtrack_*.size() > 0)
                        rowP = 1 - rowP;
                        LSYNC();
                    }
                    WSYNC();
                    if(PJ_FINE && B.dbg) { tSub[0] += tC1 - tC0; tSub[1] += clock64() - tC1; tSub[2]++; tSub[3] += cnt; }
                    if(stop) break;
                    a = b + 1;
                }
                }
            }
            chunked = !PL::LONG && !par && windowed && defCount == nDef;
            if(runLevels) { }
            else if(chunked) {
                // Column-sequential form on LDS-staged CHUNKS of the window.  Allele-rich stretches have no single-node level for
                // hundreds of levels (no cuts for the segment-parallel form) and tens to hundreds of nodes per level (the whole window
                // does not fit the staging arrays): walking them level by level against HBM cost two dependent round trips per level,
                // 13x the time of a backbone chain on the Graph M workload.  Here runs of consecutive levels whose target nodes and
                // in-edges fit the staging arrays are loaded with coalesced reads (one round trip per chunk) and the levels of a chunk
                // are solved out of LDS; the back pointers go to the wave's slab as before (window-relative CSR edge index).
                // segStart[i] = first in-edge (window-relative) of the target nodes of level i, i = 0 .. nDef
                // (with it, in the same round trip: whether a level can be solved edge-parallel, DevGraph::level_fast -> bit 25 of colInfo)
                staged_rows<7>(lane, nDef + 1, 64, [&](int i) { int2 r; r.x = G.in_off[nodeBase + P.sLev[i + 1]]; r.y = (!PL::LONG && i < nDef) ? (int)G.level_fast[level0 + i + 1] : 0; return r; },
                               [&](int i, int2 r) { P.segStart[i] = (unsigned short)(r.x - eBase); if(!PL::LONG && i < nDef && r.y) P.colInfo[i] |= (u32)r.y << 25; });
                WSYNC();
                if constexpr (!PL::LONG) if(deferRethread) {
                    // The level loop below is bound by the latency of one wave per level (its time per level does not depend on how many waves share the CU),
                    // so what it needs is MORE WAVES -- and this kernel's 11.7 KB of LDS and 143 VGPRs allow 12 per CU.  A chain whose levels all fit the
                    // staging arrays of k_rethread_chains goes there (5 KB, half the registers): its columns go out as they are (the graph character of the seed
                    // stays in seed_g for now), colInfo rides in its seed_edge slots, the window's level offsets and first in-edges in its (still unused)
                    // ext_level / ext_edge rows, the window's scalars in its DP item slots.
                    int mxN = nb - nodeBase, mxE = 0;
                    for(int i = lane; i < nDef; i += 64) { mxN = max(mxN, (int)P.sLev[i + 2] - (int)P.sLev[i + 1]); mxE = max(mxE, (int)P.segStart[i + 1] - (int)P.segStart[i]); }
                    mxN = wave_max_i32(mxN); mxE = wave_max_i32(mxE);
                    if(mxN <= RT_SN && mxE <= RT_CE) {
                        deferred = true;
                        const size_t cb = row_base(B, c);
                        for(int j = lane; j < n1; j += 64) { B.seed_level[cb + j] = LC::get(P.lvl[cur][j], lvBase); B.seed_g[cb + j] = P.g[cur][j]; B.seed_s[cb + j] = P.s[cur][j]; }
                        for(int i = lane; i < nDef; i += 64) B.seed_edge[cb + i] = (int)P.colInfo[i];
                        unsigned short* lvRow = (unsigned short*)(B.ext_level + cb); unsigned short* sgRow = (unsigned short*)(B.ext_edge + cb);
                        for(int i = lane; i <= nDef + 1; i += 64) lvRow[i] = P.sLev[i];
                        for(int i = lane; i <= nDef; i += 64) sgRow[i] = P.segStart[i];
                        if(lane == 0) {
                            B.seed_ncols[c] = n1; B.seed_begin[c] = P.startRaw; B.seed_end[c] = P.stopRaw; B.seed_removed[c] = removed; B.seed_status[c] = CHAIN_RETHREAD_PENDING;
                            int* st = (int*)B.dp_items + (size_t)c * 16;
                            *(int4*)st = make_int4(level0, nDef, nodeBase, nb); st[4] = eBase; st[5] = n1;
                        }
                    }
                }
                const int nbR = nb - nodeBase;
                int a = deferred ? nDef : 0;
                while(a < nDef && PJ_OK()) {
                    const int cand = a + lane;
                    // (chunked form of the LDS layouts: at most 62 levels, whose offsets ride in the lanes of a register)
                    const bool fits = cand < nDef && (PL::LONG || lane < 62) && ((int)P.sLev[cand + 2] - (int)P.sLev[a + 1]) <= PL::SN && ((int)P.segStart[cand + 1] - (int)P.segStart[a]) <= PL::CE;
                    const u64 fm = __ballot(fits);
                    const int cnt = (fm == ~0ull) ? 64 : (__ffsll((long long)~fm) - 1);        // levels a .. a + cnt - 1 fit together (a prefix: both sums grow)
                    if(cnt == 0) {
                        // one level wider than the staging arrays: straight from HBM
                        if(lane == 0) P.sChoice[nChunks] = (unsigned short)(a | 0x8000);
                        nChunks++;
                        const u32 ci = P.colInfo[a];
                        const unsigned char sc = (unsigned char)((ci >> 16) & 0xFFu); const bool seedIsMatch = ((ci >> 24) & 1u) != 0;
                        const int tb = nodeBase + P.sLev[a + 1], tm = (int)P.sLev[a + 2] - (int)P.sLev[a + 1], fb = nodeBase + P.sLev[a];
                        if(tm > PROJ_NODES) { if(lane == 0) PJ_FAIL(HLALA_CHAIN_ERR_FRONTIER); break; }
                        int anyReached = 0;
                        for(int z = lane; z < tm; z += 64) {
                            const int node = tb + z;
                            int best = -1, bestE = -1, bestFrom = -1;
                            const int e0 = G.in_off[node], e1 = G.in_off[node + 1];
                            for(int e = e0; e < e1; e++) {
                                const int fz = G.in_from[e] - fb; const int sp = P.Srow[rowP][fz];
                                if(sp < 0) continue;
                                const unsigned char lab = G.in_label[e];
                                if(seedIsMatch && lab != sc) continue;
                                const int cd = sp + (lab == sc ? 1 : 0);
                                if(cd > best) { best = cd; bestE = e - eBase; bestFrom = fz; }
                            }
                            edgesTouched += (u64)(e1 - e0);
                            P.Srow[1 - rowP][z] = (short)best;
                            ChoiceRec cr; cr.eid = bestE; cr.fromz = (short)bestFrom; cr.S = (short)best;
                            ch[node - nb] = cr;
                            if(best >= 0) anyReached = 1;
                        }
                        if(!__ballot(anyReached)) { if(lane == 0) PJ_FAIL(HLALA_CHAIN_ERR_INPUT); break; }
                        rowP = 1 - rowP;
                        LSYNC();
                        a++;
                        continue;
                    }
                    const long long tC0 = B.dbg ? clock64() : 0;
                    const int b = a + cnt - 1;
                    if(lane == 0) P.sChoice[nChunks] = (unsigned short)a;
                    nChunks++;
                    const int tBase = P.sLev[a + 1], nT = (int)P.sLev[b + 2] - tBase, eC = P.segStart[a], nE = (int)P.segStart[b + 1] - eC;
                    staged_rows<7>(lane, nT + 1, 64, [&](int t) { return G.in_off[nodeBase + tBase + t]; }, [&](int t, int v) { P.sIn[t] = (unsigned short)(v - eBase - eC); });
                    if constexpr (PL::LONG) {
                    staged_rows<9>(lane, nE, 64, [&](int e) { FromLab r; r.from = G.in_from[eBase + eC + e]; r.lab = G.in_label[eBase + eC + e]; return r; },
                                   [&](int e, FromLab r) { P.sFrom[e] = (unsigned short)(r.from - nodeBase); P.sLab[e] = r.lab; });
                    } else staged_rows<9>(lane, nE, 64, [&](int e) { return G.in_rec[eBase + eC + e]; }, [&](int e, u32 r) { P.eRec[e] = r; });
                    WSYNC();
                    const long long tC1 = B.dbg ? clock64() : 0;
                    bool stop = false;
                    if constexpr (PL::LONG) {
                    for(int i = a; i <= b; i++) {
                        const u32 ci = P.colInfo[i];
                        const unsigned char sc = (unsigned char)((ci >> 16) & 0xFFu); const bool seedIsMatch = ((ci >> 24) & 1u) != 0;
                        const int t0 = (int)P.sLev[i + 1] - tBase, tm = (int)P.sLev[i + 2] - (int)P.sLev[i + 1], fb = P.sLev[i];
                        if(tm > PROJ_NODES) { if(lane == 0) PJ_FAIL(HLALA_CHAIN_ERR_FRONTIER); stop = true; break; }
                        int anyReached = 0;
                        for(int z = lane; z < tm; z += 64) {
                            const int t = t0 + z;
                            int best = -1, bestE = -1, bestFrom = -1;
                            const int e0 = P.sIn[t], e1 = P.sIn[t + 1];
                            for(int e = e0; e < e1; e++) {                                   // in-edges in creation order: first maximum = smallest edge
                                const int fz = (int)P.sFrom[e] - fb; const int sp = P.Srow[rowP][fz];
                                if(sp < 0) continue;
                                const unsigned char lab = P.sLab[e];
                                if(seedIsMatch && lab != sc) continue;                        // :2803-2809
                                const int cd = sp + (lab == sc ? 1 : 0);
                                if(cd > best) { best = cd; bestE = eC + e; bestFrom = fz; }
                            }
                            edgesTouched += (u64)(e1 - e0);
                            P.Srow[1 - rowP][z] = (short)best;
                            ChoiceRec cr; cr.eid = bestE; cr.fromz = (short)bestFrom; cr.S = (short)best;
                            ch[tBase + t - nbR] = cr;
                            if(best >= 0) anyReached = 1;
                        }
                        if(!__ballot(anyReached)) { if(lane == 0) PJ_FAIL(HLALA_CHAIN_ERR_INPUT); stop = true; break; }  // assert(seedChain_backtrack_*.size() > 0)
                        rowP = 1 - rowP;
                        LSYNC();
                    }
                    WSYNC();
                    } else {
                    // The levels of the chunk, one after the other, with the lanes on the level's IN-EDGES (DevGraph::in_rec, staged above), not on its nodes.
                    // What bounds this loop is the latency of ONE wave per level (the time per level does not depend on how many waves share the CU:
                    // profiles/r03_experiments.txt), i.e. the dependent LDS round trips and branches of a level, so a level is cut down to one:
                    //   - a lane's edge fetches the score of its from-node from the LANE that computed it one level earlier (ds_bpermute; the place of that
                    //     lane is a static property of the graph and part of the edge's record) -- from the LDS row only after a level solved the other way;
                    //   - the candidates of the (at most eight) in-edges of a node sit in neighbouring lanes and are combined with DPP shifts (as many as the level's largest in-degree - 1): the key
                    //     orders by score, then by the EARLIER edge (first maximum = smallest edge, as the node-by-node loop takes it), and carries the from-node;
                    //   - the lane of a node's last in-edge holds the node's score, writes it to the row (for whoever needs it there) and the back pointer to the slab;
                    //   - the level's offsets, first in-edges, read characters and "can be solved this way" flags (DevGraph::level_fast) ride in the lanes
                    //     of four registers (readlane: no LDS round trip), the next level's records are requested a level ahead.
                    // A level with more than 64 in-edges takes overlapping slices (below); one with a node without in-edges or with more than eight is solved node by
                    // node as before (3 % of the levels of Graph M's gene windows).
                    const int lvReg = lane <= cnt + 1 ? (int)P.sLev[a + lane] : 0;
                    const int sgReg = lane <= cnt ? (int)P.segStart[a + lane] - eC : 0;
                    const u32 ciReg = lane < cnt ? P.colInfo[a + lane] : 0u;
                    bool prevFast = false; int sReg = -1;
                    u32 recN = 0;
                    { const int nE0 = __builtin_amdgcn_readlane(sgReg, 1) - __builtin_amdgcn_readlane(sgReg, 0); if(lane < nE0 && nE0 <= 64) recN = P.eRec[__builtin_amdgcn_readlane(sgReg, 0) + lane]; }
                    for(int k = 0; k < cnt; k++) {
                        const u32 ci = (u32)__builtin_amdgcn_readlane((int)ciReg, k);
                        const int sc = (int)((ci >> 16) & 0xFFu); const bool seedIsMatch = ((ci >> 24) & 1u) != 0;
                        const int l1 = __builtin_amdgcn_readlane(lvReg, k + 1), tm = __builtin_amdgcn_readlane(lvReg, k + 2) - l1;
                        const int eL0 = __builtin_amdgcn_readlane(sgReg, k), eL1 = __builtin_amdgcn_readlane(sgReg, k + 1), nEl = eL1 - eL0;
                        const int mode = (int)((ci >> 25) & 3u), maxd = (int)((ci >> 27) & 7u);
                        const u32 rec = recN;
                        if(k + 1 < cnt) { const int nEn = __builtin_amdgcn_readlane(sgReg, k + 2) - eL1; recN = (lane < nEn && nEn <= 64) ? P.eRec[eL1 + lane] : 0u; }
                        if(tm > PROJ_NODES) { if(lane == 0) PJ_FAIL(HLALA_CHAIN_ERR_FRONTIER); stop = true; break; }
                        bool reached = false;
                        // candidate of the lane's edge -> key; the keys of a node's in-edges (neighbouring lanes) -> the key of its best edge, in the lane of its last one
                        auto edge_key = [&](const u32 r, const int sp, const bool mine) -> int {
                            const int m = (int)((r >> 18) & 0xFFu) == sc ? 1 : 0;
                            const int cand = (mine && sp >= 0 && (m || !seedIsMatch)) ? sp + m + 1 : 0;       // 0: not admitted (:2803-2809) or from-node unreachable; else score + 1
                            return (cand << 15) | ((63 - lane) << 9) | (int)(r & 511u);
                        };
                        auto node_key = [&](const u32 r, const int key) -> int {
                            const int pos = (int)((r >> 15) & 7u);
                            int bk = key, kj = key;
                            for(int j = 1; j <= maxd; j++) { kj = __builtin_amdgcn_update_dpp(0, kj, 0x138, 0xF, 0xF, false); if(pos >= j) bk = max(bk, kj); }      // wave_shr:1 -- the edge j places before
                            return bk;
                        };
                        if(mode == 1) {
                            const bool mine = lane < nEl;
                            int sp;
                            if(prevFast) sp = __builtin_amdgcn_ds_bpermute((int)(((rec >> 9) & 63u) << 2), sReg);
                            else sp = (int)P.Srow[rowP][rec & 511u];
                            const int bk = node_key(rec, edge_key(rec, sp, mine));
                            const bool last = mine && (rec & (1u << 28)) != 0;
                            const u64 lastMask = __ballot(last);
                            const int best = (bk >> 15) - 1;
                            sReg = best;
                            if(last) {
                                const int tz = (int)__builtin_amdgcn_mbcnt_hi((u32)(lastMask >> 32), __builtin_amdgcn_mbcnt_lo((u32)lastMask, 0u));      // the node's rank in its level
                                ChoiceRec cr; cr.eid = best >= 0 ? eC + eL0 + 63 - ((bk >> 9) & 63) : -1; cr.fromz = (short)(best >= 0 ? (bk & 511) : -1); cr.S = (short)best;
                                P.Srow[1 - rowP][tz] = (short)best;
                                ch[l1 + tz - nbR] = cr;
                                reached = best >= 0;
                            }
                            if(lane == 0) edgesTouched += (u64)nEl;
                            prevFast = true;
                        } else if(mode == 2) {
                            // more than 64 in-edges: slices of 64 that overlap by seven lanes, so that a node whose last in-edge lies in a slice's lanes 7 .. 63 has all
                            // its (at most eight) in-edges in that slice; the lanes 0 .. 6 of a later slice only provide them
                            int nodesDone = 0;
                            for(int s0 = 0; s0 == 0 || s0 + 7 < nEl; s0 += 57) {
                                const bool mine = s0 + lane < nEl;
                                const u32 r = mine ? P.eRec[eL0 + s0 + lane] : 0u;
                                const int sp = (int)P.Srow[rowP][r & 511u];
                                const int bk = node_key(r, edge_key(r, sp, mine));
                                const bool last = mine && (r & (1u << 28)) != 0 && (s0 == 0 || lane >= 7);
                                const u64 lastMask = __ballot(last);
                                if(last) {
                                    const int best = (bk >> 15) - 1;
                                    const int tz = nodesDone + (int)__builtin_amdgcn_mbcnt_hi((u32)(lastMask >> 32), __builtin_amdgcn_mbcnt_lo((u32)lastMask, 0u));
                                    ChoiceRec cr; cr.eid = best >= 0 ? eC + eL0 + s0 + 63 - ((bk >> 9) & 63) : -1; cr.fromz = (short)(best >= 0 ? (bk & 511) : -1); cr.S = (short)best;
                                    P.Srow[1 - rowP][tz] = (short)best;
                                    ch[l1 + tz - nbR] = cr;
                                    if(best >= 0) reached = true;
                                }
                                nodesDone += (int)__popcll(lastMask);
                            }
                            if(lane == 0) edgesTouched += (u64)nEl;
                            prevFast = false;
                        } else {
                            const int t0 = l1 - tBase;
                            for(int z = lane; z < tm; z += 64) {
                                const int t = t0 + z;
                                int best = -1, bestE = -1, bestFrom = -1;
                                const int e0 = P.sIn[t], e1 = P.sIn[t + 1];
                                for(int e = e0; e < e1; e++) {                               // in-edges in creation order: first maximum = smallest edge
                                    const u32 r = P.eRec[e];
                                    const int fz = (int)(r & 511u);
                                    const int sp = P.Srow[rowP][fz];
                                    if(sp < 0) continue;
                                    const int lab = (int)((r >> 18) & 0xFFu);
                                    if(seedIsMatch && lab != sc) continue;                    // :2803-2809
                                    const int cd = sp + (lab == sc ? 1 : 0);
                                    if(cd > best) { best = cd; bestE = eC + e; bestFrom = fz; }
                                }
                                edgesTouched += (u64)(e1 - e0);
                                P.Srow[1 - rowP][z] = (short)best;
                                ChoiceRec cr; cr.eid = bestE; cr.fromz = (short)bestFrom; cr.S = (short)best;
                                ch[tBase + t - nbR] = cr;
                                if(best >= 0) reached = true;
                            }
                            prevFast = false;
                        }
                        if(!__ballot(reached)) { if(lane == 0) PJ_FAIL(HLALA_CHAIN_ERR_INPUT); stop = true; break; }  // assert(seedChain_backtrack_*.size() > 0)
                        rowP = 1 - rowP;
                        LSYNC();
                    }
                    WSYNC();
                    }
                    if(B.dbg) { tSub[0] += tC1 - tC0; tSub[1] += clock64() - tC1; tSub[2]++; tSub[3] += cnt; }
                    if(stop) break;
                    a = b + 1;
                }
            } else if(!par) {
                for(int j = 0; j < n1; j++) {
                    int l = uni(LC::get(P.lvl[cur][j], lvBase));
                    if(l == -1) continue;                                                     // :2710-2714
                    unsigned char sc = P.s[cur][j], gc = P.g[cur][j];
                    bool seedIsMatch = (sc == gc);
                    int tb = G.level_off[l + 1], tm = G.level_off[l + 2] - tb, fb = G.level_off[l];
                    if(tm > PROJ_NODES) { if(lane == 0) PJ_FAIL(HLALA_CHAIN_ERR_FRONTIER); break; }
                    int anyReached = 0;
                    for(int z = lane; z < tm; z += 64) {
                        int node = tb + z;
                        int best = -1, bestE = -1, bestFrom = -1;
                        int e0 = G.in_off[node], e1 = G.in_off[node + 1];
                        for(int e = e0; e < e1; e++) {
                            int fz = G.in_from[e] - fb; int sp = P.Srow[rowP][fz];
                            if(sp < 0) continue;
                            unsigned char lab = G.in_label[e];
                            if(seedIsMatch && lab != sc) continue;
                            int cand = sp + (lab == sc ? 1 : 0);
                            if(cand > best) { best = cand; bestE = G.in_eid[e]; bestFrom = fz; }
                        }
                        edgesTouched += (u64)(e1 - e0);
                        P.Srow[1 - rowP][z] = (short)best;
                        ChoiceRec cr; cr.eid = bestE; cr.fromz = (short)bestFrom; cr.S = (short)best;
                        ch[node - nb] = cr;
                        if(best >= 0) anyReached = 1;
                    }
                    if(!__ballot(anyReached)) { if(lane == 0) PJ_FAIL(HLALA_CHAIN_ERR_INPUT); break; }  // assert(seedChain_backtrack_*.size() > 0)
                    rowP = 1 - rowP;
                    LSYNC();                      // (the score rows are LDS; the back pointers go to HBM and are not read before the backtrace)
                }
            }
            WSYNC();
            if constexpr (PL::LONG && PJ_FINE) if(B.dbg) { dbgN[2] = n1; dbgN[3] = nDef; dbgN[4] = chCount; dbgN[5] = nSeg; dbgN[6] = nLong; dbgN[7] = (par ? 1 : 0) | (fastLong ? 2 : 0) | (staged ? 4 : 0) | (windowed ? 8 : 0); }
            PJ_T(5); PJ_F(5);
            // ---------------- backtrace (:2838-3007)
            if(PJ_OK() && !deferred) {
                int lastLevel = uni(LC::get(P.lvl[cur][n1 - 1], lvBase));
                int tb, tm; const short* lastS;
                if(par && !lastLong) { tb = nodeBase + P.sLev[nDef]; tm = P.sLev[nDef + 1] - P.sLev[nDef]; lastS = (const short*)(Sflat + P.sLev[nDef]); }      // (lastLong: the window's last levels were solved level by level -- their scores are in the row)
                else { tb = G.level_off[lastLevel + 1]; tm = G.level_off[lastLevel + 2] - tb; lastS = P.Srow[rowP]; }
                int bestS = -1;
                for(int z = lane; z < tm; z += 64) bestS = max(bestS, (int)lastS[z]);
                bestS = wave_max_i32(bestS);
                int zsel = 0x7FFFFFFF;
                for(int z = lane; z < tm; z += 64) if(lastS[z] == bestS) { zsel = min(zsel, z); }
                zsel = -wave_max_i32(-zsel);                                                   // *(runningN.begin()): smallest node among the maxima (:2867)
                const size_t cb = row_base(B, c);
                if(par || fastLong) {
                    // every segment is traced from its right cut node (the last one from the selected node); the other column
                    // buffer takes the chosen edge per column (window-relative: CSR in-edge - eBase), then all lanes emit (edge ids are independent HBM reads)
                    auto pick = P.lvl[1 - cur];
                    for(int j = lane; j < n1; j += 64) pick[j] = LC::pickNone();
                    WSYNC(); PJ_F(6);
                    const int nbR = nb - nodeBase;
                    if(par) for(int sg = lane; sg < nSeg; sg += 64) {
                        const int a = P.segStart[sg], b = P.segStart[sg + 1];
                        if(PL::LONG && b - a > PROJ_SEGMAX) continue;
                        int t = (b == nDef) ? (int)P.sLev[nDef] + zsel : (int)P.sLev[b];
                        for(int i = b - 1; i >= a; i--) { int e = P.sChoice[t - nbR]; pick[P.colInfo[i] & 0xFFFFu] = e; t = P.sFrom[e]; }
                    }
                    if constexpr (PL::LONG) {
                    // the ranges solved level by level: their chunks again, last to first (P.chunkStart) -- the back pointers of a chunk's nodes are staged into LDS with coalesced
                    // reads and lane 0 follows them there.  A range that ends the window starts from the selected node, any other from its cut node.
                    const u32* const chw = (const u32*)slabCh;
                    for(int rq = nLong - 1; rq >= 0; rq--) {
                    const int ra = uni(P.longSeg[2 * rq]), rb = uni(P.longSeg[2 * rq + 1]);
                    int z = rb == nDef ? zsel : 0;
                    for(int bb = rb - 1; bb >= ra; ) {
                        int w = bb >> 6; u64 m = __hip_atomic_load(P.chunkStart.p + w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & (~0ull >> (63 - (bb & 63)));
                        while(m == 0) { w--; m = __hip_atomic_load(P.chunkStart.p + w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }      // (the range's first level starts a chunk)
                        const int aa = w * 64 + 63 - __clzll((long long)m);
                        const int i0 = aa + lane;
                        const int lvA = i0 <= nDef + 1 ? G.level_off[level0 + i0] : 0;
                        const int lvB = (i0 <= bb && i0 + 1 <= nDef + 1) ? G.level_off[level0 + i0 + 1] : 0;
                        const u32 ciReg = i0 <= bb ? (u32)P.colInfo[i0] : 0u;
                        const int sgA = i0 <= bb ? G.in_off[lvB] - eBase : 0;
                        const int tBase = __builtin_amdgcn_readlane(lvA, 1), nT = __builtin_amdgcn_readlane(lvA, bb - aa + 2) - tBase;
                        {
                            // (a chunk holds at most RT_SN nodes, a level on its own at most PROJ_NODES = RT_CE)
                            for(int t0 = 0; t0 < nT; t0 += 4 * 64) {
                                u32 vr[4];
                                #pragma unroll
                                for(int u = 0; u < 4; u++) { const int t = t0 + lane + 64 * u; if(t < nT) vr[u] = chw[tBase + t - nb]; }
                                #pragma unroll
                                for(int u = 0; u < 4; u++) { const int t = t0 + lane + 64 * u; if(t < nT) P.cRec[t] = vr[u]; }
                            }
                        }
                        WSYNC();
                        for(int i = bb; i >= aa; i--) {
                            const int col = (int)((u32)__builtin_amdgcn_readlane((int)ciReg, i - aa) & 0xFFFFu), lvI = __builtin_amdgcn_readlane(lvA, i - aa + 1), sgI = __builtin_amdgcn_readlane(sgA, i - aa);
                            if(lane == 0) { const u32 r = P.cRec[lvI - tBase + z]; pick[col] = sgI + (int)(r >> 16); z = (int)(short)(r & 0xFFFFu); }
                        }
                        WSYNC();
                        bb = aa - 1;
                    }
                    }
                    }
                    WSYNC(); PJ_F(7);
                    if constexpr (PL::LONG) {
                    staged_rows<PJL_U>(lane, n1, 64, [&](int j) { const int pe = pick[j]; FromLab r; r.from = -1; r.lab = '_';
                                                              if(pe >= 0) { r.from = G.in_eid[eBase + pe]; r.lab = G.in_label[eBase + pe]; } return r; },
                        [&](int j, FromLab r) {
                            B.seed_level[cb + j] = r.from < 0 ? -1 : LC::get(P.lvl[cur][j], lvBase); B.seed_edge[cb + j] = r.from; B.seed_g[cb + j] = r.lab;
                            B.seed_s[cb + j] = P.s[cur][j];
                        });
                    } else {
                    staged_rows<6>(lane, n1, 64, [&](int j) { const typename PL::LvT pe = pick[j]; return LC::pickIsNone(pe) ? -1 : G.in_eid[eBase + (int)pe]; },
                        [&](int j, int eid) {
                            const typename PL::LvT pe = pick[j];
                            if(LC::pickIsNone(pe)) { B.seed_level[cb + j] = -1; B.seed_edge[cb + j] = -1; B.seed_g[cb + j] = '_'; }
                            else { B.seed_level[cb + j] = LC::get(P.lvl[cur][j], lvBase); B.seed_edge[cb + j] = eid; B.seed_g[cb + j] = P.sLab[(int)pe]; }
                            B.seed_s[cb + j] = P.s[cur][j];
                        });
                    }
                } else if(chunked) {
                    // the chunks again, last to first: the back pointers of a chunk's nodes are staged into LDS with coalesced reads, lane 0
                    // follows them there, then all lanes emit
                    auto pick = P.lvl[1 - cur];
                    for(int j = lane; j < n1; j += 64) pick[j] = LC::pickNone();
                    WSYNC();
                    const int nbR = nb - nodeBase;
                    int z = zsel;
                    for(int k = nChunks - 1; k >= 0; k--) {
                        const int a = P.sChoice[k] & 0x7FFF; const bool wide = (P.sChoice[k] & 0x8000) != 0;
                        const int b = (k + 1 < nChunks ? (int)(P.sChoice[k + 1] & 0x7FFF) : nDef) - 1;
                        if(wide) {
                            if(lane == 0) { const ChoiceRec cr = ch[(int)P.sLev[a + 1] + z - nbR]; pick[P.colInfo[a] & 0xFFFFu] = cr.eid; z = cr.fromz; }
                        } else {
                            const int tBase = P.sLev[a + 1], nT = (int)P.sLev[b + 2] - tBase;
                            staged_rows<7>(lane, nT, 64, [&](int t) { return ch[tBase + t - nbR]; }, [&](int t, ChoiceRec cr) { P.sIn[t] = (unsigned short)cr.fromz; P.sFrom[t] = (unsigned short)cr.eid; });
                            WSYNC();
                            if(lane == 0) for(int i = b; i >= a; i--) { const int t = (int)P.sLev[i + 1] - tBase + z; pick[P.colInfo[i] & 0xFFFFu] = (int)P.sFrom[t]; z = (int)(short)P.sIn[t]; }
                        }
                        WSYNC();
                    }
                    staged_rows<6>(lane, n1, 64, [&](int j) { const typename PL::LvT pe = pick[j]; FromLab r; r.from = -1; r.lab = '_';
                                                              if(!LC::pickIsNone(pe)) { r.from = G.in_eid[eBase + (int)pe]; r.lab = G.in_label[eBase + (int)pe]; } return r; },
                        [&](int j, FromLab r) {
                            B.seed_level[cb + j] = LC::pickIsNone(pick[j]) ? -1 : LC::get(P.lvl[cur][j], lvBase); B.seed_edge[cb + j] = r.from; B.seed_g[cb + j] = r.lab;
                            B.seed_s[cb + j] = P.s[cur][j];
                        });
                } else if(lane == 0) {
                    int node = tb + zsel;
                    for(int j = n1 - 1; j >= 0; j--) {
                        int l = LC::get(P.lvl[cur][j], lvBase);
                        if(l == -1) { B.seed_level[cb + j] = -1; B.seed_edge[cb + j] = -1; B.seed_g[cb + j] = '_'; B.seed_s[cb + j] = P.s[cur][j]; continue; }
                        ChoiceRec cr = ch[node - nb];
                        B.seed_level[cb + j] = l; B.seed_edge[cb + j] = cr.eid; B.seed_g[cb + j] = G.edge_label[cr.eid]; B.seed_s[cb + j] = P.s[cur][j];
                        node = G.level_off[l] + cr.fromz;
                    }
                }
                if(lane == 0) {
                    B.seed_ncols[c] = n1; B.seed_begin[c] = P.startRaw; B.seed_end[c] = P.stopRaw; B.seed_removed[c] = removed;
                    accCols += (u64)n1;
                }
            }
        }
        PJ_T(6); PJ_F(8);
        if constexpr (PL::LONG && PJ_FINE) readOrd++;
        if(B.dbg) { for(int i = 0; i < 6; i++) if(tPh[i + 1] && tPh[i]) tAcc[i] += tPh[i + 1] - tPh[i]; tAcc[6]++; }
        if constexpr (PL::LONG && PJ_FINE) if(B.dbg && lane == 0 && tPh[6] && tPh[0]) {       // (HLALA_DEBUG=1) histogram of a chain's cycles by their binary logarithm: counts at dbg[4096 ..], sums (cycles >> 16) at dbg[4160 ..] -- tools/long_phase.py
            const long long cyc = tPh[6] - tPh[0]; int k = 0; while(k < 47 && (cyc >> (k + 1)) != 0) k++;
            atomicAdd(&B.dbg[4096 + k], 1); atomicAdd(&B.dbg[4160 + k], (int)(cyc >> 16));
            { const int o = readOrd < 31 ? readOrd : 31; atomicAdd(&B.dbg[4400 + o], (int)(cyc >> 16)); atomicAdd(&B.dbg[4432 + o], 1); atomicAdd(&B.dbg[4464 + o], dbgN[2] >> 4); }      // by the read's ordinal on its wavefront
            if(k >= 25) {       // the heaviest reads on their own: phases at dbg[4320 ..], the pieces of tFine at dbg[4330 ..] (cycles >> 12), their number at dbg[4319]
                { const int q = atomicAdd(&B.dbg[4318], 1); if(q < 300) { int* r = B.dbg + 5000 + 10 * q; r[0] = c; r[1] = dbgN[0]; r[2] = dbgN[1]; r[3] = dbgN[2]; r[4] = dbgN[3]; r[5] = dbgN[4]; r[6] = dbgN[5]; r[7] = dbgN[6]; r[8] = (int)(cyc >> 16); r[9] = dbgN[7] | ((int)((tPh[0] >> 20) & 0x7FFFF) << 8); } }
                atomicAdd(&B.dbg[4319], 1);
                for(int i = 0; i < 6; i++) atomicAdd(&B.dbg[4320 + i], (int)((tPh[i + 1] - tPh[i]) >> 12));
                for(int i = 0; i < 12; i++) atomicAdd(&B.dbg[4330 + i], (int)((tFine[i] - tFine0[i]) >> 12));
                for(int i = 0; i < 4; i++) atomicAdd(&B.dbg[4344 + i], (int)((tSub[i] - tSub0[i]) >> (i < 2 ? 12 : 0)));
            }
        }
        {
            int e = wave_sum_i32((int)edgesTouched);
            accEdges += (u64)e;
        }
        WSYNC();
        if(!PJ_OK() && lane == 0) { B.seed_status[c] = P.err; B.seed_ncols[c] = 0; }
        }   // chain survives the filters
        WSYNC();
        }
    }
    if(lane == 0) { if(accCols) atomicAdd(&B.counters[CNT_SEED_COLS], accCols); if(accEdges) atomicAdd(&B.counters[CNT_EDGES], accEdges); }
#ifndef HLALA_DP_TIMING      // (the timing build of the DP kernels puts k_stitch_chains' clocks into the same counters)
    if(B.dbg && lane == 0) { for(int i = 0; i < 6; i++) atomicAdd(&B.counters[16 + i], (u64)tAcc[i]); atomicAdd(&B.counters[23], (u64)tAcc[6]); }
    if(B.dbg && lane == 0) for(int i = 0; i < 4; i++) atomicAdd(&B.counters[24 + i], (u64)tSub[i]);
    if constexpr (PL::LONG && PJ_FINE) if(B.dbg && lane == 0) for(int i = 0; i < 12; i++) atomicAdd(&B.dbg[4300 + i], (int)(tFine[i] >> 12));
#endif
}

// ---------------------------------------------------------------------------------------------------------------------------------------
// k_rethread_chains -- the chunked form of the re-threading DP (processBAM.cpp:2676-3007) for the chains k_project_chains left pending.
// The level loop of that form is bound by the latency of ONE wave per level -- a dependent chain of ~120 instructions; its time per level is the
// same with 2 and with 14 waves on a CU (profiles/r03_experiments.txt) -- so throughput is waves per CU, and k_project_chains, which carries the
// whole projection (11.7 KB of LDS, 143 VGPRs), tops out at 12.  This kernel holds only what the loop needs: a chunk's in-edge records and in-edge
// offsets, two score rows, the picks of the backtrace (5.3 KB); a chunk's level offsets, first in-edges and colInfo ride in the lanes of registers
// (a chunk has at most 62 levels), read from the rows k_project_chains left in HBM.  Same recurrence, same order of the ties (first maximum = smallest
// edge), same end node (smallest among the maxima), same chunking rule per chain as the wave-wide form there; see the comments at that loop.
struct __align__(16) RethreadLds {
    u32 eRec[RT_CE];                        // DevGraph::in_rec of the chunk's in-edges; backtrace: from-node (low half) | chosen edge (high half) of the chunk's nodes
    unsigned short sIn[RT_SN + 2];          // in_off of the chunk's nodes (levels solved node by node)
    short Srow[2][PROJ_NODES];
    unsigned short pick[PROJ_CAP];          // per column: chosen in-edge (window-relative), 0xFFFF = none
    u64 chunkStart[PROJ_CAP / 64];          // levels at which a chunk started
};

__global__ __launch_bounds__(64, 6) void k_rethread_chains(const DevGraph* __restrict__ Gp, const DevBatch* __restrict__ Bp, char* slabs, size_t slabBytes)
{
    const DevGraph& G = *Gp;
    const DevBatch& B = *Bp;
    __shared__ RethreadLds P;
    const int lane = lane_id();
    // back pointer of a window node: from-node rank (low half, 0xFFFF = none) | chosen in-edge, window-relative (high half) -- ONE word per node (the 8-byte
    // ChoiceRec of k_project_chains carries the score as well, which the backtrace never reads: half of this kernel's slab traffic)
    u32* const ch = (u32*)(slabs + (size_t)blockIdx.x * slabBytes);
    const int stride = B.stride;
    u64 accCols = 0, accEdges = 0;
    const int nWork = ordered_chains(B);
    for(;;) {
        // RT_DRAW chains per draw: the wave takes the pending ones among them, one after the other.  (64 per draw until the chains came in position
        // order: the pending chains -- gene windows -- then sit together in the list, a draw holds up to 64 of them at 300 k cycles each, and the waves
        // that drew the last full ones finished 10 ms after the others.)
        constexpr int RT_DRAW = 16;
        int c0 = 0;
        if(lane == 0) c0 = atomicAdd(&B.work_counter[3], RT_DRAW);
        c0 = __builtin_amdgcn_readfirstlane(c0);
        if(c0 >= nWork) break;
        const int cq = (lane < RT_DRAW && c0 + lane < nWork) ? (B.chain_order ? B.chain_order[c0 + lane] : c0 + lane) : -1;        // position order, as in k_project_chains
        u64 pend = __ballot(cq >= 0 && B.seed_status[cq] == CHAIN_RETHREAD_PENDING);
        for(; pend; pend &= pend - 1) {
            const int c = __builtin_amdgcn_readlane(cq, __ffsll((long long)pend) - 1);
            const size_t cb = row_base(B, c);
            const int* st = (const int*)B.dp_items + (size_t)c * 16;
            int sv = lane < 6 ? st[lane] : 0;
            const int level0 = __builtin_amdgcn_readlane(sv, 0), nDef = __builtin_amdgcn_readlane(sv, 1), nodeBase = __builtin_amdgcn_readlane(sv, 2), nb = __builtin_amdgcn_readlane(sv, 3),
                      eBase = __builtin_amdgcn_readlane(sv, 4), n1 = __builtin_amdgcn_readlane(sv, 5);
            (void)level0;
            const unsigned short* lvRow = (const unsigned short*)(B.ext_level + cb); const unsigned short* sgRow = (const unsigned short*)(B.ext_edge + cb);
            const int* ciRow = B.seed_edge + cb;
            const int nbR = nb - nodeBase;
            for(int z = lane; z < nbR; z += 64) P.Srow[0][z] = 0;                          // all nodes of the first level, S = 0 (:2694-2701)
            if(lane < PROJ_CAP / 64) P.chunkStart[lane] = 0;
            WSYNC();
            int rowP = 0, err = 0;
            u64 edgesTouched = 0;
            int a = 0;
            while(a < nDef && !err) {
                // the chunk's levels: offsets a .. a + 63 (+ 2), first in-edges a .. a + 63 (+ 1), colInfo -- one round trip
                const int i0 = a + lane;
                const int lvA = i0 <= nDef + 1 ? (int)lvRow[i0] : 0, lvC = i0 + 2 <= nDef + 1 ? (int)lvRow[i0 + 2] : 0;
                const int sgA = i0 <= nDef ? (int)sgRow[i0] : 0, sgB = i0 + 1 <= nDef ? (int)sgRow[i0 + 1] : 0;
                const u32 ciReg = i0 < nDef ? (u32)ciRow[i0] : 0u;
                const int lvA1 = __builtin_amdgcn_readlane(lvA, 1), sgA0 = __builtin_amdgcn_readlane(sgA, 0);
                const bool fits = i0 < nDef && lane < 62 && (lvC - lvA1) <= RT_SN && (sgB - sgA0) <= RT_CE;
                const u64 fm = __ballot(fits);
                const int cnt = __ffsll((long long)~fm) - 1;                               // levels a .. a + cnt - 1 fit together (a prefix: both sums grow)
                if(cnt == 0) { err = HLALA_CHAIN_ERR_FRONTIER; break; }                    // (k_project_chains keeps every chain with a level beyond these arrays)
                const int b = a + cnt - 1;
                if(lane == 0) P.chunkStart[a >> 6] |= 1ull << (a & 63);
                const int tBase = lvA1, nT = __builtin_amdgcn_readlane(lvA, cnt + 1) - tBase, eC = sgA0, nE = __builtin_amdgcn_readlane(sgA, cnt) - eC;
                const int lvReg = lvA, sgReg = sgA - eC;
                {   // the chunk's in-edge records and in-edge offsets: one round trip
                    constexpr int UE = RT_CE / 64, UN = (RT_SN + 1 + 63) / 64;
                    // (the in-edge offsets of the nodes are only read by the levels that are solved node by node -- mode 0, 3 % of the levels: most chunks have none)
                    const bool needOff = __ballot(lane < cnt && ((ciReg >> 25) & 3u) == 0u) != 0;
                    u32 ve[UE]; int vn[UN];
                    #pragma unroll
                    for(int u = 0; u < UE; u++) { const int e = lane + 64 * u; if(e < nE) ve[u] = G.in_rec[eBase + eC + e]; }
                    if(needOff) {
                        #pragma unroll
                        for(int u = 0; u < UN; u++) { const int t = lane + 64 * u; if(t <= nT) vn[u] = G.in_off[nodeBase + tBase + t]; }
                    }
                    #pragma unroll
                    for(int u = 0; u < UE; u++) { const int e = lane + 64 * u; if(e < nE) P.eRec[e] = ve[u]; }
                    if(needOff) {
                        #pragma unroll
                        for(int u = 0; u < UN; u++) { const int t = lane + 64 * u; if(t <= nT) P.sIn[t] = (unsigned short)(vn[u] - eBase - eC); }
                    }
                }
                WSYNC();
                bool prevFast = false; int sReg = -1;
                u32 recN = 0;
                { const int nE0 = __builtin_amdgcn_readlane(sgReg, 1); if(lane < nE0 && nE0 <= 64) recN = P.eRec[lane]; }
                for(int k = 0; k < cnt; k++) {
                    const u32 ci = (u32)__builtin_amdgcn_readlane((int)ciReg, k);
                    const int sc = (int)((ci >> 16) & 0xFFu); const bool seedIsMatch = ((ci >> 24) & 1u) != 0;
                    const int l1 = __builtin_amdgcn_readlane(lvReg, k + 1), tm = __builtin_amdgcn_readlane(lvReg, k + 2) - l1;
                    const int eL0 = __builtin_amdgcn_readlane(sgReg, k), eL1 = __builtin_amdgcn_readlane(sgReg, k + 1), nEl = eL1 - eL0;
                    const int mode = (int)((ci >> 25) & 3u), maxd = (int)((ci >> 27) & 7u);
                    const u32 rec = recN;
                    if(k + 1 < cnt) { const int nEn = __builtin_amdgcn_readlane(sgReg, k + 2) - eL1; recN = (lane < nEn && nEn <= 64) ? P.eRec[eL1 + lane] : 0u; }
                    bool reached = false;
                    auto edge_key = [&](const u32 r, const int sp, const bool mine) -> int {
                        const int m = (int)((r >> 18) & 0xFFu) == sc ? 1 : 0;
                        const int cand = (mine && sp >= 0 && (m || !seedIsMatch)) ? sp + m + 1 : 0;           // 0: not admitted (:2803-2809) or from-node unreachable; else score + 1
                        return (cand << 15) | ((63 - lane) << 9) | (int)(r & 511u);
                    };
                    auto node_key = [&](const u32 r, const int key) -> int {
                        const int pos = (int)((r >> 15) & 7u);
                        int bk = key, kj = key;
                        for(int j = 1; j <= maxd; j++) { kj = __builtin_amdgcn_update_dpp(0, kj, 0x138, 0xF, 0xF, false); if(pos >= j) bk = max(bk, kj); }          // wave_shr:1 -- the edge j places before
                        return bk;
                    };
                    if(mode == 1) {
                        const bool mine = lane < nEl;
                        int sp;
                        if(prevFast) sp = __builtin_amdgcn_ds_bpermute((int)(((rec >> 9) & 63u) << 2), sReg);
                        else sp = (int)P.Srow[rowP][rec & 511u];
                        const int bk = node_key(rec, edge_key(rec, sp, mine));
                        const bool last = mine && (rec & (1u << 28)) != 0;
                        const u64 lastMask = __ballot(last);
                        const int best = (bk >> 15) - 1;
                        sReg = best;
                        if(last) {
                            const int tz = (int)__builtin_amdgcn_mbcnt_hi((u32)(lastMask >> 32), __builtin_amdgcn_mbcnt_lo((u32)lastMask, 0u));
                            P.Srow[1 - rowP][tz] = (short)best;
                            ch[l1 + tz - nbR] = best >= 0 ? ((u32)(bk & 511) | ((u32)(eC + eL0 + 63 - ((bk >> 9) & 63)) << 16)) : 0xFFFFFFFFu;
                            reached = best >= 0;
                        }
                        if(lane == 0) edgesTouched += (u64)nEl;
                        prevFast = true;
                    } else if(mode == 2) {
                        int nodesDone = 0;
                        for(int s0 = 0; s0 == 0 || s0 + 7 < nEl; s0 += 57) {
                            const bool mine = s0 + lane < nEl;
                            const u32 r = mine ? P.eRec[eL0 + s0 + lane] : 0u;
                            const int sp = (int)P.Srow[rowP][r & 511u];
                            const int bk = node_key(r, edge_key(r, sp, mine));
                            const bool last = mine && (r & (1u << 28)) != 0 && (s0 == 0 || lane >= 7);
                            const u64 lastMask = __ballot(last);
                            if(last) {
                                const int best = (bk >> 15) - 1;
                                const int tz = nodesDone + (int)__builtin_amdgcn_mbcnt_hi((u32)(lastMask >> 32), __builtin_amdgcn_mbcnt_lo((u32)lastMask, 0u));
                                P.Srow[1 - rowP][tz] = (short)best;
                                ch[l1 + tz - nbR] = best >= 0 ? ((u32)(bk & 511) | ((u32)(eC + eL0 + s0 + 63 - ((bk >> 9) & 63)) << 16)) : 0xFFFFFFFFu;
                                if(best >= 0) reached = true;
                            }
                            nodesDone += (int)__popcll(lastMask);
                        }
                        if(lane == 0) edgesTouched += (u64)nEl;
                        prevFast = false;
                    } else {
                        const int t0 = l1 - tBase;
                        for(int z = lane; z < tm; z += 64) {
                            const int t = t0 + z;
                            int best = -1, bestE = -1, bestFrom = -1;
                            const int e0 = P.sIn[t], e1 = P.sIn[t + 1];
                            for(int e = e0; e < e1; e++) {                                   // in-edges in creation order: first maximum = smallest edge
                                const u32 r = P.eRec[e];
                                const int fz = (int)(r & 511u);
                                const int sp = P.Srow[rowP][fz];
                                if(sp < 0) continue;
                                const int lab = (int)((r >> 18) & 0xFFu);
                                if(seedIsMatch && lab != sc) continue;                        // :2803-2809
                                const int cd = sp + (lab == sc ? 1 : 0);
                                if(cd > best) { best = cd; bestE = eC + e; bestFrom = fz; }
                            }
                            edgesTouched += (u64)(e1 - e0);
                            P.Srow[1 - rowP][z] = (short)best;
                            ch[tBase + t - nbR] = (u32)(unsigned short)(short)bestFrom | ((u32)(unsigned short)bestE << 16);
                            if(best >= 0) reached = true;
                        }
                        prevFast = false;
                    }
                    if(!__ballot(reached)) { err = HLALA_CHAIN_ERR_INPUT; break; }          // assert(seedChain_backtrack_*.size() > 0)
                    rowP = 1 - rowP;
                    LSYNC();          // (round 6: the lanes exchange scores through LDS and lane permutes only; WSYNC() also waited for the level's back-pointer stores to reach HBM)
                }
                WSYNC();
                a = b + 1;
            }
            // ---- backtrace (:2838-3007)
            if(!err) {
                const int lvLast = (int)lvRow[nDef], tmL = (int)lvRow[nDef + 1] - lvLast;
                const short* lastS = P.Srow[rowP];
                int bestS = -1;
                for(int z = lane; z < tmL; z += 64) bestS = max(bestS, (int)lastS[z]);
                bestS = wave_max_i32(bestS);
                int zsel = 0x7FFFFFFF;
                for(int z = lane; z < tmL; z += 64) if(lastS[z] == bestS) zsel = min(zsel, z);
                zsel = -wave_max_i32(-zsel);                                               // *(runningN.begin()): smallest node among the maxima (:2867)
                for(int j = lane; j < n1; j += 64) P.pick[j] = 0xFFFF;
                WSYNC();
                int z = zsel;
                for(int bb = nDef - 1; bb >= 0; ) {
                    int w = bb >> 6; u64 m = P.chunkStart[w] & (~0ull >> (63 - (bb & 63)));
                    while(m == 0) { w--; m = P.chunkStart[w]; }                            // (level 0 starts a chunk)
                    const int aa = w * 64 + 63 - __clzll((long long)m);
                    const int i0 = aa + lane;
                    const int lvA = i0 <= nDef + 1 ? (int)lvRow[i0] : 0;
                    const u32 ciReg = i0 <= bb ? (u32)ciRow[i0] : 0u;
                    const int tBase = __builtin_amdgcn_readlane(lvA, 1), nT = __builtin_amdgcn_readlane(lvA, bb - aa + 2) - tBase;
                    {   // the back pointers of the chunk's nodes: from-node | chosen edge per node, one round trip
                        constexpr int UN = (RT_SN + 63) / 64;
                        u32 vr[UN];
                        #pragma unroll
                        for(int u = 0; u < UN; u++) { const int t = lane + 64 * u; if(t < nT) vr[u] = ch[tBase + t - nbR]; }
                        #pragma unroll
                        for(int u = 0; u < UN; u++) { const int t = lane + 64 * u; if(t < nT) P.eRec[t] = vr[u]; }
                    }
                    WSYNC();
                    for(int i = bb; i >= aa; i--) {
                        const int col = (int)((u32)__builtin_amdgcn_readlane((int)ciReg, i - aa) & 0xFFFFu), lvI = __builtin_amdgcn_readlane(lvA, i - aa + 1);
                        if(lane == 0) { const u32 r = P.eRec[lvI - tBase + z]; P.pick[col] = (unsigned short)(r >> 16); z = (int)(short)(r & 0xFFFFu); }
                    }
                    WSYNC();
                    bb = aa - 1;
                }
                staged_rows<6>(lane, n1, 64, [&](int j) { const unsigned short pe = P.pick[j]; FromLab r; r.from = -1; r.lab = '_';
                                                          if(pe != 0xFFFF) { r.from = G.in_eid[eBase + pe]; r.lab = G.in_label[eBase + pe]; } return r; },
                               [&](int j, FromLab r) { B.seed_edge[cb + j] = r.from; B.seed_g[cb + j] = r.lab; });
                if(lane == 0) { B.seed_status[c] = HLALA_CHAIN_OK; accCols += (u64)n1; }
            } else if(lane == 0) { B.seed_status[c] = err; B.seed_ncols[c] = 0; }
            accEdges += edgesTouched;
            WSYNC();
        }
    }
    accCols = wave_sum_u64(accCols); accEdges = wave_sum_u64(accEdges);
    if(lane == 0) { if(accCols) atomicAdd(&B.counters[CNT_SEED_COLS], accCols); if(accEdges) atomicAdd(&B.counters[CNT_EDGES], accEdges); }
}

}  // namespace hlala
