// kernel_dp.hip -- stage B: extensionAligner::extendSeedChain + scoreOneAlignment on gfx950.
//
// The work item of the extension stage is ONE call of the affine X-drop frontier DP
// fullNeedleman_diagonal_extension_gapJumper (mapper/aligner/extensionAligner.cpp:335-1556): the left or the
// right extension of one seed chain (extendSeedChain, :220-319).  A frontier of this DP holds a handful of
// cells (median 3-4 on 2x150 bp pairs), so a DP is run by a GROUP of lanes, not by a whole wavefront:
//
//   DpTiny   16 lanes per DP, 4 DPs per wavefront   frontier <= 16 cells, <= 24 candidate targets per iteration
//   DpMid    32 lanes per DP, 2 DPs per wavefront   frontier <= 32, <= 96 targets        (DPs that outgrow DpTiny)
//   DpSmall  64 lanes per DP                        frontier <= 64, <= 96 targets        (DPs that outgrow DpMid)
//   DpWide   64 lanes per DP                        frontier <= 256, <= 384 targets
//   DpBroad / DpLarge / DpHuge   one DP per BLOCK of two wavefronts (DP_TAIL_THREADS)   frontier <= 512 / 1024 / 8192 (the last with its state in HBM)
// A DP that outgrows its class is queued for the first later class that holds what overflowed and runs there again; every class computes identical results
// (one template).  Groups of up to one wavefront synchronise with wave fences, blocks of several wavefronts with s_barrier, and the group collectives below
// exchange one word per wave through LDS.
//
// Every group is a small state machine (fetch -> iterate ... -> select end cell -> backtrace -> expand -> done);
// the four groups of a wavefront advance independently inside one persistent loop, so a DP that ends early
// immediately makes room for the next item and the slow serial part of one DP (the back-pointer chase) is
// spread over the iterations of the other three.  Group collectives stay on the DPP cross-lane path: a group
// of 16 is exactly one DPP row.
//
// Per iteration (identical in all classes):
//   generate : one lane per frontier cell pushes its candidates into an LDS hash keyed by the target cell; ties
//              are resolved with ds_max_u32 on (score, reversed push index), which is exactly the reference's
//              "first maximum in push order" (Utilities.cpp:379-406)
//   evaluate : one lane per target cell combines the three matrices, applies the -16 keep threshold, merges into
//              the cell table (HBM scratch slab private to the group) and derives the running-maximum / patience
//              bookkeeping with group reductions
//   filter   : X-drop window of 15 below the iteration maximum, then a rank sort by (x,y,z) so the next
//              iteration pushes in the reference's std::map order.
// Scores are integers (the reference's doubles only ever hold integers, alignerBase.cpp:19-25).
//
// k_dp_items      : per chain, input checks of extendSeedChain and the DP items, one slot per chain in position order (kernel_order.hip); counts per block and list
// k_dp_lists      : the ten dense item lists of the first class (band x 3, jump-free, general; left / right each) from the scanned counts
// k_dp<C, TIER>   : the DP classes above (the first one in two instantiations: DpTinyJF for calls that meet no gap-path jump, DpTiny for the rest); extension
//                   columns go straight into the chain's output row
// k_stitch_chains : extendWithOtherSeedChain / extendToFullSequenceLength (verboseSeedChain.cpp:23-136) and
//                   scoreOneAlignment (extensionAligner.cpp:52-182), one wavefront per chain
#include "batch.h"
#include "../../include/hlala_gpu.h"

namespace hlala {

enum { K_DIAG = 0, K_GGAP = 1, K_SGAP = 2, K_HOP = 3, K_JUMP = 4 };
enum { M_D = 0, M_GG = 1, M_SG = 2 };
enum { PH_IDLE = 0, PH_RUN, PH_SELECT, PH_BT, PH_EXPAND, PH_DONE };

constexpr u64 HKEY_EMPTY = ~0ull;
#ifndef HLALA_DP_PROFILE_LOG2
#define HLALA_DP_PROFILE_LOG2 18      // (profile build: calls of more than 2^this cycles are recorded)
#endif
#ifndef HLALA_EARLY_GEN_MAX
#define HLALA_EARLY_GEN_MAX 0xFFFFFE      // (tools/gpu_gen_wrap.sh builds with a tiny value to exercise the wrap-around under the parity tests)
#endif
constexpr int DP_EARLY_GEN_MAX = HLALA_EARLY_GEN_MAX;      // generations of a slab's early table before it is cleared again (24 bits of the entry word)
constexpr int EXT_PENDING = 0x7FFFFFFF;     // ext_status of a chain whose DP items are in flight
constexpr int DP_BT_STEPS_PER_TRIP = 32;    // back pointers one group follows per trip of the persistent loop (measured: 3 -> 236 ms, 6 -> 231, 12 -> 226, 24 -> 225, 64 -> 224 per 524 k pairs)

// (16-lane class: a target table of 32 entries, i.e. at most 24 targets per iteration, sends 2.6 % more calls on to the 32-lane class than 64 entries did and
// makes the kernel 10 % faster: half the table to scan, reset and keep in LDS; 8-lane groups were tried for this class -- 8 DPs per wave -- and lost)
// The three classes of the long tail (frontiers of 257 .. 8192 cells: a few thousand DP calls per million pairs, each hundreds of iterations over hundreds of cells)
// run one DP per BLOCK of several wavefronts: the rounds of an iteration over the frontier / the targets are split across the waves, block barriers take the
// place of the wave fences and the group collectives exchange one word per wave through LDS.  (One wavefront per DP left most of the chip idle while a few
// hundred waves walked their frontiers 64 cells at a time: 120 ms of side stream per million pairs.)  HLALA_DP_TAIL_THREADS = 64 restores one wave per DP.
// Measured on Graph M, 1 M pairs per batch (tools/gpu_tail_variants.sh; ms per step with two batches in flight / one batch at a time; tail alone):
//   64 threads 281 / 336 (tail 120 ms)   128 threads 280 / 306 (86 ms)   256 threads 299 / 300 (80 ms)   in-memory class at 1024 threads: 315 / 345 (spills)
// -- four waves per DP shorten the tail most but take issue slots from the main-stream kernels of the next batch (16-lane class 91 -> 118 ms beside them).
#ifndef HLALA_DP_TAIL_THREADS
#define HLALA_DP_TAIL_THREADS 128
#endif
constexpr int DP_TAIL_THREADS = HLALA_DP_TAIL_THREADS;
#ifndef HLALA_DP_TINY_WAVES
#define HLALA_DP_TINY_WAVES 4          // waves per SIMD the 16-lane kernel is compiled for (registers: 512 / waves)
#endif
#ifndef HLALA_DP_WIDE_THREADS
#define HLALA_DP_WIDE_THREADS 64
#endif
#ifndef HLALA_DP_BROAD_THREADS
#define HLALA_DP_BROAD_THREADS HLALA_DP_TAIL_THREADS
#endif
#ifndef HLALA_DP_LARGE_THREADS
#define HLALA_DP_LARGE_THREADS HLALA_DP_TAIL_THREADS
#endif
#ifndef HLALA_DP_HUGE_THREADS
#define HLALA_DP_HUGE_THREADS HLALA_DP_TAIL_THREADS
#endif
struct DpTiny  { static constexpr bool JF = false; static constexpr int THREADS = 64, WAVES = HLALA_DP_TINY_WAVES, GW = 16, WCAP = 16,   HC = 32,   IBITS = 4,  CELLS = 2048,     EARLY = 4096,     IMPCAP = 64,   COMPLETED = 256,          STEPS = 1024;     typedef u32 Best; typedef short Slot; typedef unsigned char ImpIdx; static constexpr bool IN_MEMORY = false; };
// The 16-lane class for DP calls that are known to meet no gap-path jump (k_dp_items: FlatGraph::jfree_out / jfree_in beyond the reach of the call -- two
// thirds of the calls of the Graph M workload).  Without jumps no cell is ever created ahead of its diagonal, so no cell is met twice: the instantiation
// is compiled WITHOUT the early-cell table, its look-ups, the second evaluate pass, the staged improvements of existing cells and the jump candidates
// (a fifth fewer instructions, no spilled register at four waves per SIMD), and -- more to the point -- its wavefronts never execute those paths because
// ONE of their four groups needs them.  A call that meets a jump after all (the bound is a heuristic, not a proof) fails over to the next class like a call that
// outgrew its capacity and is re-run there: results do not depend on the classification.
#ifndef HLALA_DP_TINYJF_WAVES
#define HLALA_DP_TINYJF_WAVES 4
#endif
struct DpTinyJF : DpTiny { static constexpr bool JF = true; static constexpr int WAVES = HLALA_DP_TINYJF_WAVES; };
struct DpMid   { static constexpr bool JF = false; static constexpr int THREADS = 64, WAVES = 4, GW = 32, WCAP = 32,   HC = 128,  IBITS = 5,  CELLS = 4096,     EARLY = 8192,     IMPCAP = 256,  COMPLETED = 512,          STEPS = 2048;     typedef u32 Best; typedef short Slot; typedef unsigned short ImpIdx; static constexpr bool IN_MEMORY = false; };
struct DpSmall { static constexpr bool JF = false; static constexpr int THREADS = 64, WAVES = 5, GW = 64, WCAP = 64,   HC = 128,  IBITS = 7,  CELLS = DP_CELLS, EARLY = DP_CELLS, IMPCAP = 2048, COMPLETED = DP_COMPLETED, STEPS = DP_STEPS; typedef u32 Best; typedef short Slot; typedef unsigned short ImpIdx; static constexpr bool IN_MEMORY = false; };
// frontiers of 65 .. 256 cells (the bulk of what outgrows the 64-lane class on allele-rich levels): one wavefront per DP like the large class,
// but a quarter of its LDS, so that seven of them share a CU instead of one
struct DpWide  { static constexpr bool JF = false; static constexpr int THREADS = HLALA_DP_WIDE_THREADS, WAVES = 2, GW = HLALA_DP_WIDE_THREADS, WCAP = 256,  HC = 512,  IBITS = 8,  CELLS = DP_CELLS, EARLY = DP_CELLS, IMPCAP = 2048, COMPLETED = DP_COMPLETED, STEPS = DP_STEPS; typedef u32 Best; typedef short Slot; typedef unsigned short ImpIdx; static constexpr bool IN_MEMORY = false; };
// frontiers of 257 .. 512 cells with the table sizes of the large class (tens of thousands of kept cells): half the LDS of the large class, three per CU
struct DpBroad { static constexpr bool JF = false; static constexpr int THREADS = HLALA_DP_BROAD_THREADS, WAVES = 1, GW = HLALA_DP_BROAD_THREADS, WCAP = 512,  HC = 1024, IBITS = 9,  CELLS = DP_CELLS_LARGE, EARLY = DP_CELLS_LARGE, IMPCAP = 4096, COMPLETED = DP_COMPLETED_LARGE, STEPS = DP_STEPS; typedef u32 Best; typedef int Slot; typedef unsigned short ImpIdx; static constexpr bool IN_MEMORY = false; };
// (allele-rich levels of a real PRG -- hundreds of nodes per level, SURVEY.md 8(d) Graph M: frontiers of 700+ cells, 16 000+ kept cells and
//  thousands of sequence-complete cells per DP were measured -- are what the large class is sized for; its table slots are ints)
struct DpLarge { static constexpr bool JF = false; static constexpr int THREADS = HLALA_DP_LARGE_THREADS, WAVES = 1, GW = HLALA_DP_LARGE_THREADS, WCAP = 1024, HC = 2048, IBITS = 10, CELLS = DP_CELLS_LARGE, EARLY = DP_CELLS_LARGE, IMPCAP = 4096, COMPLETED = DP_COMPLETED_LARGE, STEPS = DP_STEPS; typedef u32 Best; typedef int Slot; typedef unsigned short ImpIdx; static constexpr bool IN_MEMORY = false; };

// The backstop: everything the other classes keep in LDS -- target table, frontiers, DP state -- lives in the block's HBM slab, so the capacities are
// set by memory, not by the 160 KB of a CU (frontiers of 3000+ cells, 60 000 kept cells and 15 000 tied complete cells per DP occur on the densest
// levels of the Graph M workload: about 30 DP calls per million pairs).  Same code (one template): the structure reference simply points into
// the slab, the wave fences become agent-scope fences (plain loads must not hit stale L1 lines of words the atomics changed in L2), and the
// frontier sort borrows the otherwise unused LDS.  An order of magnitude slower per cell than the LDS classes; nothing is dropped.
struct DpHuge  { static constexpr bool JF = false; static constexpr int THREADS = HLALA_DP_HUGE_THREADS, WAVES = 1, GW = HLALA_DP_HUGE_THREADS, WCAP = 8192, HC = 16384, IBITS = 13, CELLS = 131072, EARLY = 131072, IMPCAP = 8192, COMPLETED = 65536, STEPS = DP_STEPS; typedef u64 Best; typedef int Slot; typedef unsigned short ImpIdx; static constexpr bool IN_MEMORY = true; };
// (key, payload) pairs of the in-memory class's frontier sort held in LDS: 64 KB.  Frontiers beyond 4096 cells -- none on the Graph M workload: the widest holds 3200 --
// are sorted in the block's slab in HBM.  (8192 pairs = 128 KB, 4096 and 2048 measured the same within 1 % with two batches in flight and alone,
// profiles/r03_experiments.txt: beside the next batch's persistent kernels the class is short of issue slots, not of LDS.  64 KB leaves a CU room for other blocks.)
#ifndef HLALA_DP_SORT_SCRATCH
#define HLALA_DP_SORT_SCRATCH 4096
#endif
constexpr int DP_SORT_SCRATCH = HLALA_DP_SORT_SCRATCH;

// State of one DP call.  It lives in the group's LDS block (all lanes of the group read the same words, a broadcast), so
// that only the phase has to stay in registers across the states of the persistent loop.
struct DpState {
    int itemIdx;                                          // index into the item list (kept for a requeue)
    int item, rOff, seqLen, start_seq, startLevel, startNode;
    int d, b1, b2, n1, n2, nCells, nCompleted, curMax, firstMaxSlot, lastInc, earlyInit, itersRun, diagonals;
    int earlyMaxNat;                                      // largest natural diagonal |dx|+|dy| among the cells that were created ahead of it
    u32 cellsEvaluated;
    int endSlot, endScore, nSteps, nCols;
    int have, sb, se, err;
    int isAlias;                                          // the results now being produced are those of a linked duplicate of the DP that ran
    int needTier;                                         // capacity failure: first tier whose class can hold what overflowed (0 = the next one)
};

template <bool SMALL> struct TlistT { typedef unsigned short type; };
template <> struct TlistT<true> { typedef unsigned char type; };

template <class C>
struct __align__(16) DpLdsT {
    u64 hkey[C::HC];
    typename C::Best hbest[3][C::HC];
    typename TlistT<(C::HC <= 256)>::type tlist[C::HC];     // hash entries in use this iteration
    // two frontier buffers: the cells of the last diagonal (b1) and of the one before (b2); the new frontier of an iteration is written over b2, whose
    // cells no candidate and no cached score needs once the targets are evaluated (a third buffer used to hold it: a third of the frontier LDS)
    u64 fkey[2][C::WCAP];
    typename C::Slot fslot[2][C::WCAP];        // table slot of the frontier cell
    short fD[2][C::WCAP], fG[2][C::WCAP], fS[2][C::WCAP];
    typename C::Slot tes[C::HC];    // per target: existing / assigned table slot (-1 = none)
    unsigned char timp[C::HC];      // per target: improved-matrix mask | 0x80 = new cell
    typename C::ImpIdx hq[C::HC];   // per hash entry: index of the improvement its cell staged this iteration (all ones = none)
    int nNew, nImp, nKeepF, err, nCompletedAdd;
    int chunkNext, chunkEnd;                          // items of the group's current draw from the item list (several per atomic: k_dp)
    int jumpMet;                                      // jump-free instantiation: the call met a gap-path jump after all (it goes on to the general 16-lane list)
    int nTa;                                          // targets claimed so far this iteration (classes of several waves per DP: the waves append to one list)
    int nextPhase;                                    // state after PH_DONE: idle, or the end-cell choice of a linked duplicate
    int btSlot, btM, btX, btY, btGuard, btDone;       // back-pointer chase in progress (lane 0 of the group)
    DpState st;
    u64 accCalls, accIters, accCells, accEdges;       // work counters of the DPs this group finished (flushed once at exit)
#ifdef HLALA_DP_TIMING
    long long tPh[16];
#endif
#ifdef HLALA_DP_PROFILE
    long long pfStart; int pfSlow, pfImp, pfPre, pfMaxNT, pfMaxF; long long pfPh[16];
#endif
};

// one kept DP cell in the slab: 32 bytes, written with two 16-byte stores
struct __align__(16) CellRec { u64 key; short sc[4]; u32 bt[3]; u32 pad; };      // sc = D, GG, SG, -; bt per matrix

// Scratch of one DP call in HBM (private to its group).  Only the base address is held in registers; every array sits at a
// compile-time offset, 8-byte arrays first.
template <class C>
struct DpSlabT {
    char* base;
    static constexpr size_t O_CELL      = 0;                                          // CellRec [CELLS]
    static constexpr size_t O_EARLY_KEY = O_CELL + (size_t)C::CELLS * 32;             // u64  [EARLY]
    static constexpr size_t O_STEP_XY   = O_EARLY_KEY + (size_t)C::EARLY * 8;         // u64  [STEPS]
    static constexpr size_t O_IMP_KEY   = O_STEP_XY + (size_t)C::STEPS * 8;           // u64  [IMPCAP]
    static constexpr size_t O_IMP_NEW   = O_IMP_KEY + (size_t)C::IMPCAP * 8;          // short[4*IMPCAP]
    static constexpr size_t O_IMP_BT    = O_IMP_NEW + (size_t)C::IMPCAP * 8;          // u32  [3*IMPCAP]
    static constexpr size_t O_STEP_BT   = O_IMP_BT + (size_t)C::IMPCAP * 12;          // u32  [STEPS]
    static constexpr size_t O_EARLY_VAL = O_STEP_BT + (size_t)C::STEPS * 4;           // int  [EARLY]
    static constexpr size_t O_COMPLETED = O_EARLY_VAL + (size_t)C::EARLY * 4;         // int  [COMPLETED]
    static constexpr size_t O_IMP_SLOT  = O_COMPLETED + (size_t)C::COMPLETED * 4;     // int  [IMPCAP]
    static constexpr size_t O_IMP_MASK  = O_IMP_SLOT + (size_t)C::IMPCAP * 4;         // int  [IMPCAP]
    static constexpr size_t O_TIE_SLOT  = O_IMP_MASK + (size_t)C::IMPCAP * 4;         // int  [COMPLETED]
    static constexpr size_t O_TIE_KEY   = (O_TIE_SLOT + (size_t)C::COMPLETED * 4 + 7) & ~(size_t)7;   // u64 [COMPLETED]
    static constexpr size_t O_EARLY_GEN = O_TIE_KEY + (size_t)C::COMPLETED * 8;         // int: generation of the early table = number of DP calls that used it (pools start zeroed)
    static constexpr size_t O_SORT_SPILL = (O_EARLY_GEN + 8 + 15) & ~(size_t)15;        // u64[2 * WCAP], in-memory class only: (key, payload) pairs of a frontier wider than the LDS scratch
    static constexpr size_t BYTES       = (O_SORT_SPILL + (C::IN_MEMORY ? (size_t)C::WCAP * 16 : 0) + 255) & ~(size_t)255;
    __device__ __forceinline__ CellRec* cell() const { return (CellRec*)(base + O_CELL); }
    __device__ __forceinline__ u64* early_key() const { return (u64*)(base + O_EARLY_KEY); }
    __device__ __forceinline__ u32* step_bt() const { return (u32*)(base + O_STEP_BT); }
    __device__ __forceinline__ u64* step_xy() const { return (u64*)(base + O_STEP_XY); }
    __device__ __forceinline__ u64* imp_key() const { return (u64*)(base + O_IMP_KEY); }
    __device__ __forceinline__ u32* imp_bt() const { return (u32*)(base + O_IMP_BT); }
    __device__ __forceinline__ short* imp_new() const { return (short*)(base + O_IMP_NEW); }
    __device__ __forceinline__ int* early_val() const { return (int*)(base + O_EARLY_VAL); }
    __device__ __forceinline__ int* completed() const { return (int*)(base + O_COMPLETED); }
    __device__ __forceinline__ int* imp_slot() const { return (int*)(base + O_IMP_SLOT); }
    __device__ __forceinline__ int* imp_mask() const { return (int*)(base + O_IMP_MASK); }
    __device__ __forceinline__ int* tie_slot() const { return (int*)(base + O_TIE_SLOT); }
    __device__ __forceinline__ u64* tie_key() const { return (u64*)(base + O_TIE_KEY); }
    __device__ __forceinline__ int* early_gen() const { return (int*)(base + O_EARLY_GEN); }
    __device__ __forceinline__ u64* sort_spill() const { return (u64*)(base + O_SORT_SPILL); }
};

template <class C>
__host__ __device__ inline size_t dp_slab_bytes() { return DpSlabT<C>::BYTES; }
// bytes of one block of an in-memory class: the structure the other classes keep in LDS, then the scratch slab
template <class C>
__host__ __device__ inline size_t dp_inmemory_bytes() { return ((sizeof(DpLdsT<C>) + 255) & ~(size_t)255) + DpSlabT<C>::BYTES; }

// ------------------------------------------------------------------------------------------ group collectives
// GW = 64: the wave-wide DPP reductions of device_common.h (results are wave-uniform, in SGPRs).
// GW = 16: one DPP row; all-reduce with quad_perm xor 1 / xor 2, row_half_mirror, row_mirror (4 VALU ops, every lane
//          of the row ends with the result).  Rows never exchange data, so groups may sit in divergent code.
template <int GW> __device__ __forceinline__ int grp_lane() { return (int)(threadIdx.x & (GW - 1)); }
template <int GW> __device__ __forceinline__ int grp_base() { return (int)(threadIdx.x & 63 & ~(GW - 1)); }

#define HLALA_ROW_ALLREDUCE(v, OP)                                                          \
    do {                                                                                    \
        int t_;                                                                             \
        t_ = __builtin_amdgcn_update_dpp(v, v, 0xB1, 0xF, 0xF, false); v = OP(v, t_);       /* quad_perm [1,0,3,2] */ \
        t_ = __builtin_amdgcn_update_dpp(v, v, 0x4E, 0xF, 0xF, false); v = OP(v, t_);       /* quad_perm [2,3,0,1] */ \
        t_ = __builtin_amdgcn_update_dpp(v, v, 0x141, 0xF, 0xF, false); v = OP(v, t_);      /* row_half_mirror */     \
        t_ = __builtin_amdgcn_update_dpp(v, v, 0x140, 0xF, 0xF, false); v = OP(v, t_);      /* row_mirror */          \
    } while(0)

// GW > 64 (one DP per block of several wavefronts): a block barrier that also orders the block's LDS / global accesses, and an exchange of one word per
// wave through LDS.  The second barrier of a collective keeps a fast wave from overwriting the words before every wave has read them.
__device__ __forceinline__ void blk_barrier()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
template <int GW> struct BlkX { static constexpr int NW = GW / 64; };
template <int GW> __device__ __forceinline__ int* blk_words() { __shared__ int x[BlkX<GW>::NW > 0 ? 2 * BlkX<GW>::NW : 2]; return x; }

template <int GW> __device__ __forceinline__ int grp_max_i32(int v)
{
    if(GW == 64) return wave_max_i32(v);
    if constexpr (GW > 64) {
        const int w = wave_max_i32(v); int* x = blk_words<GW>();
        if(lane_id() == 0) x[threadIdx.x >> 6] = w;
        blk_barrier();
        int r = x[0];
#pragma unroll
        for(int i = 1; i < BlkX<GW>::NW; i++) r = op_max_(r, x[i]);
        blk_barrier();
        return r;
    } else {
    HLALA_ROW_ALLREDUCE(v, op_max_);
    if(GW == 32) v = op_max_(v, __shfl_xor(v, 16));     // the partner row of a 32-lane group
    return v;
    }
}
template <int GW> __device__ __forceinline__ int grp_sum_i32(int v)
{
    if(GW == 64) return wave_sum_i32(v);
    if constexpr (GW > 64) {
        const int w = wave_sum_i32(v); int* x = blk_words<GW>();
        if(lane_id() == 0) x[threadIdx.x >> 6] = w;
        blk_barrier();
        int r = x[0];
#pragma unroll
        for(int i = 1; i < BlkX<GW>::NW; i++) r += x[i];
        blk_barrier();
        return r;
    } else {
    HLALA_ROW_ALLREDUCE(v, op_add_);
    if(GW == 32) v = op_add_(v, __shfl_xor(v, 16));
    return v;
    }
}
template <int GW> __device__ __forceinline__ u64 grp_min_u64(u64 v)
{
    if(GW == 64) return wave_min_u64(v);
    int hi = (int)(~(u32)(v >> 32) ^ 0x80000000u);
    int mh = grp_max_i32<GW>(hi);
    int lo = (hi == mh) ? (int)(~(u32)v ^ 0x80000000u) : (int)0x80000000;
    int ml = grp_max_i32<GW>(lo);
    return ((u64)(~((u32)mh ^ 0x80000000u)) << 32) | (u64)(~((u32)ml ^ 0x80000000u));
}
// exclusive prefix sum over the group (v >= 0); `total` is group-uniform
template <int GW> __device__ __forceinline__ int grp_excl_scan(int v, int& total)
{
    if(GW == 64) return wave_excl_scan(v, total);
    if constexpr (GW > 64) {
        int wt; const int off = wave_excl_scan(v, wt); int* x = blk_words<GW>();
        const int wv = (int)(threadIdx.x >> 6);
        if(lane_id() == 0) x[wv] = wt;
        blk_barrier();
        int before = 0, tot = 0;
#pragma unroll
        for(int i = 0; i < BlkX<GW>::NW; i++) { const int t = x[i]; if(i < wv) before += t; tot += t; }
        blk_barrier();
        total = tot;
        return before + off;
    } else {
    int x = v, t;
    t = dpp_mov<0x111>(0, x); x += t;       // row_shr:1 (lanes without a source keep 0)
    t = dpp_mov<0x112>(0, x); x += t;
    t = dpp_mov<0x114>(0, x); x += t;
    t = dpp_mov<0x118>(0, x); x += t;
    int m = x; HLALA_ROW_ALLREDUCE(m, op_max_);     // inclusive sums are non-decreasing: the maximum is the row total
    if(GW == 32) {
        const int other = __shfl_xor(m, 16);
        if(threadIdx.x & 16) x += other;             // second row of the group continues after the first
        m += other;
    }
    total = m;
    return x - v;
    }
}
template <int GW> __device__ __forceinline__ u64 grp_ballot(bool p)      // groups of at most one wavefront
{
    static_assert(GW <= 64, "grp_ballot: one wavefront at most (grp_any / grp_rank for blocks)");
    u64 b = __ballot(p);
    if(GW == 64) return b;
    return (b >> grp_base<GW>()) & ((1ull << (GW & 63)) - 1ull);
}
// does the predicate hold for any lane of the group
template <int GW> __device__ __forceinline__ bool grp_any(bool p)
{
    if constexpr (GW > 64) {
        const int w = __ballot(p) != 0 ? 1 : 0; int* x = blk_words<GW>();
        if(lane_id() == 0) x[threadIdx.x >> 6] = w;
        blk_barrier();
        int r = 0;
#pragma unroll
        for(int i = 0; i < BlkX<GW>::NW; i++) r |= x[i];
        blk_barrier();
        return r != 0;
    } else return grp_ballot<GW>(p) != 0;
}
// number of lanes before this one (in group order) for which the predicate holds; `total` = how many in the group (group-uniform)
template <int GW> __device__ __forceinline__ int grp_rank(bool p, int& total)
{
    if constexpr (GW > 64) {
        const u64 m = __ballot(p);
        const int off = (int)__builtin_amdgcn_mbcnt_hi((u32)(m >> 32), __builtin_amdgcn_mbcnt_lo((u32)m, 0u));
        int* x = blk_words<GW>(); const int wv = (int)(threadIdx.x >> 6);
        if(lane_id() == 0) x[wv] = __popcll(m);
        blk_barrier();
        int before = 0, tot = 0;
#pragma unroll
        for(int i = 0; i < BlkX<GW>::NW; i++) { const int t = x[i]; if(i < wv) before += t; tot += t; }
        blk_barrier();
        total = tot;
        return before + off;
    } else {
        const u64 m = grp_ballot<GW>(p);
        total = __popcll(m);
        return __popcll(m & ((1ull << grp_lane<GW>()) - 1ull));
    }
}
// the value lane 0 of the group holds, in every lane
template <int GW> __device__ __forceinline__ int grp_bcast0(int v)
{
    if(GW == 64) return __builtin_amdgcn_readfirstlane(v);
    if constexpr (GW > 64) {
        int* x = blk_words<GW>();
        if(threadIdx.x == 0) x[0] = v;
        blk_barrier();
        const int r = x[0];
        blk_barrier();
        return __builtin_amdgcn_readfirstlane(r);
    } else return __shfl(v, grp_base<GW>());
}
// a group-uniform value: scalar for groups of whole wavefronts, left alone otherwise
template <int GW> __device__ __forceinline__ int guni(int v) { return GW >= 64 ? __builtin_amdgcn_readfirstlane(v) : v; }

// ordering between the lanes of a DP's group: a wavefront-scope fence for state in LDS; for the in-memory class an agent-scope fence (L1
// invalidate / write-back around the barrier), because its atomics execute in the L2 and plain loads of the same words would otherwise be
// served from the CU's L1
template <class C> __device__ __forceinline__ void dp_sync()
{
    // (round 5: the RELEASE is workgroup-scope.  A DP call's state is written and read by the wavefronts of ONE block, i.e. one CU: its stores are through the CU's L1 and in the
    //  XCD's L2 once they are counted done, which is all a reader on the same CU needs -- what it must not do is hit a stale L1 line of a word an L2 atomic changed, hence the
    //  agent-scope ACQUIRE (L1 invalidate).  The agent-scope release this used to be is an L2 WRITE-BACK of the whole XCD's dirty lines, a dozen times per iteration of a call:
    //  it slowed the class and every kernel beside it -- profiles/r05_experiments.txt 12, 13.)
    //  The workgroup-scope release is only enough while the wavefronts of a block share one L1, i.e. NOT in threadgroup-split mode (-mtgsplit): refuse that build;
    //  make EXTRA=-DHLALA_DP_AGENT_RELEASE keeps the agent-scope release of rounds 2-4 (hlala_build_flags() reports it; tools/gpu_r6_agentrel.sh runs the parity suite on it).
#if defined(__AMDGCN_TGSPLIT__) || defined(__gfx950_tgsplit__)
#error "kernel_dp.hip: dp_sync releases at workgroup scope -- build with -DHLALA_DP_AGENT_RELEASE for threadgroup-split mode"
#endif
#ifdef HLALA_DP_AGENT_RELEASE
#define DP_RELEASE_SCOPE "agent"
#else
#define DP_RELEASE_SCOPE "workgroup"
#endif
    if constexpr (C::IN_MEMORY) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, DP_RELEASE_SCOPE); if constexpr (C::GW > 64) __builtin_amdgcn_s_barrier(); else __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); }
    else if constexpr (C::GW > 64) blk_barrier();
    else { WSYNC(); }
}
#define DSYNC() dp_sync<C>()

// DP cell key: level x (24 bits) | read offset y (12 bits) | node id (28 bits).  Node ids are level-major and
// stable in creation order, so unsigned key order == the reference's std::map order (x, then y, then rank z).
__device__ __forceinline__ u64 mk_key(int x, int y, int node) { return ((u64)(u32)x << 40) | ((u64)(u32)y << 28) | (u64)(u32)node; }
__device__ __forceinline__ int key_x(u64 k) { return (int)(k >> 40); }
__device__ __forceinline__ int key_y(u64 k) { return (int)((k >> 28) & 0xFFF); }
__device__ __forceinline__ int key_node(u64 k) { return (int)(k & 0xFFFFFFF); }
// targets of one iteration differ in a few low bits of node / y / x: node + 17 y + 31 x spreads them over neighbouring
// entries without a full-width multiply (v_mul_lo_u32 is quarter rate; shifts, adds and the 24-bit mad are full rate)
__device__ __forceinline__ u32 hash64(u64 k)
{
    const u32 node = (u32)k & 0xFFFFFFFu, y = (u32)(k >> 28) & 0xFFFu, x = (u32)(k >> 40);
    u32 h = node + __umul24(y, 17u) + __umul24(x, 31u);
    return h ^ (h >> 6) ^ (h >> 12);
}

// The per-DP cell hash in the slab holds up to every kept cell of a DP (tens of thousands in allele-rich gap regions) and is probed
// linearly: it needs a hash that scatters.  hash64 maps the cells of one region onto a narrow range of values (fine for the per-iteration
// LDS hash, whose few hundred keys form short runs), which made the probe chains of this table thousands of entries long -- one DP call
// of the Graph M workload took 1.3 s in compare-and-swaps.  Two multiplies per call, only in iterations that can meet an early cell.
__device__ __forceinline__ u32 hash_mix(u64 k)
{
    u32 h = (u32)k * 0x9E3779B1u ^ (u32)(k >> 32) * 0x85EBCA77u;
    h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 12;
    return h;
}

// hash of the per-iteration LDS target table: the cheap one for the small classes (a few dozen keys in a table twice that size); the
// large class fills its 2048 entries half and more on allele-rich levels, where the narrow value range of hash64 turned linear probing
// into walks of hundreds of compare-and-swaps per claim
template <class C> __device__ __forceinline__ u32 tgt_hash(u64 k) { return (C::HC > 256 ? hash_mix(k) : hash64(k)) & (u32)(C::HC - 1); }

// entry of a key in the per-iteration target hash, -1 if it is not a target of this iteration (no insertion; entries are never removed
// while an iteration is evaluated)
template <class C>
__device__ inline int dp_find(const DpLdsT<C>& S, u64 key)
{
    u32 h = tgt_hash<C>(key);
#pragma nounroll
    for(int probe = 0; probe < C::HC; probe++) {
        const u64 cur = S.hkey[h];
        if(cur == key) return (int)h;
        if(cur == HKEY_EMPTY) return -1;
        h = (h + 1) & (C::HC - 1);
    }
    return -1;
}

// back pointer: previous cell slot (17 bits, CELLS <= 131072) | source matrix (2) | kind (3) | local push index j (8; 0xFF = none)
__device__ __forceinline__ u32 mk_bt(int prev, int src, int kind, int edge) { return (u32)prev | ((u32)src << 17) | ((u32)kind << 19) | (((u32)edge & 0xFFu) << 22); }
__device__ __forceinline__ int bt_prev(u32 b) { return (int)(b & 0x1FFFF); }
__device__ __forceinline__ int bt_src(u32 b) { return (int)((b >> 17) & 3); }
__device__ __forceinline__ int bt_kind(u32 b) { return (int)((b >> 19) & 7); }
__device__ __forceinline__ int bt_edge(u32 b) { return (int)((b >> 22) & 0xFF); }


// candidate value: (score, reversed push index) so that an unsigned max = highest score, earliest push.  The push index has IBITS + 9 bits (frontier
// entry, phase, edge / jump); every LDS class packs it with the score (+64: < 2^13 for reads up to DP_SEQCAP) into 32 bits -- half the LDS of the
// value arrays and 32-bit ds_max for the wide classes, which used to carry 64-bit values --, the in-memory class keeps 64.
template <class C> struct BestBits { static constexpr int OB = (C::IBITS + 9 <= 16) ? 16 : C::IBITS + 9; static_assert(sizeof(typename C::Best) == 8 || OB <= 19, "score field too narrow"); };
template <class C> __device__ __forceinline__ void pack_best(u32& o, int score, int order) { constexpr int OB = BestBits<C>::OB; o = ((u32)(score + 64) << OB) | (u32)(((1 << OB) - 1) - order); }
template <class C> __device__ __forceinline__ void pack_best(u64& o, int score, int order) { o = ((u64)(u32)(score + 64) << 32) | (u64)(u32)(0x7FFFFFFF - order); }
template <class C> __device__ __forceinline__ int best_score(u32 b) { return b ? (int)(b >> BestBits<C>::OB) - 64 : DP_NEG; }
template <class C> __device__ __forceinline__ int best_score(u64 b) { return b ? (int)(b >> 32) - 64 : DP_NEG; }
template <class C> __device__ __forceinline__ int best_order(u32 b) { constexpr int OB = BestBits<C>::OB; return ((1 << OB) - 1) - (int)(b & (u32)((1 << OB) - 1)); }
template <class C> __device__ __forceinline__ int best_order(u64 b) { return 0x7FFFFFFF - (int)(b & 0xFFFFFFFFull); }

// push one candidate (Alt::{D,GG,SG}.push_back in the reference) -- returns false on hash overflow.
// One LDS round trip per probe: the compare-and-swap both claims an empty entry and reports the resident key.
template <class C>
__device__ inline bool dp_push(DpLdsT<C>& S, u64 key, int mat, int score, int order, int& claimedAt)      // claimedAt: the entry, if this push took a free one
{
    u32 h = tgt_hash<C>(key);
#pragma nounroll
    for(int probe = 0; probe < C::HC; probe++) {
        u64 old = atomicCAS(&S.hkey[h], HKEY_EMPTY, key);
        if(old == HKEY_EMPTY) claimedAt = (int)h;
        if(old == HKEY_EMPTY || old == key) { typename C::Best v; pack_best<C>(v, score, order); atomicMax(&S.hbest[mat][h], v); return true; }
        h = (h + 1) & (C::HC - 1);
    }
    return false;
}

// (the classes that collect their target list from the table afterwards have no use for the claim)
template <class C>
__device__ inline bool dp_push(DpLdsT<C>& S, u64 key, int mat, int score, int order)
{
    u32 h = tgt_hash<C>(key);
#pragma nounroll
    for(int probe = 0; probe < C::HC; probe++) {
        u64 old = atomicCAS(&S.hkey[h], HKEY_EMPTY, key);
        if(old == HKEY_EMPTY || old == key) { typename C::Best v; pack_best<C>(v, score, order); atomicMax(&S.hbest[mat][h], v); return true; }
        h = (h + 1) & (C::HC - 1);
    }
    return false;
}

// continue the probe sequence of a claim that found a different key at its home entry; returns HC when the hash is full
template <class C>
__device__ inline u32 dp_probe(DpLdsT<C>& S, u64 key, u32 h)
{
#pragma nounroll
    for(int probe = 1; probe < C::HC; probe++) {
        h = (h + 1) & (C::HC - 1);
        u64 old = atomicCAS(&S.hkey[h], HKEY_EMPTY, key);
        if(old == HKEY_EMPTY || old == key) return h;
    }
    return (u32)C::HC;
}
// ... reporting whether the entry was free (claim-time target list)
template <class C>
__device__ inline u32 dp_probe(DpLdsT<C>& S, u64 key, u32 h, bool& claimed)
{
#pragma nounroll
    for(int probe = 1; probe < C::HC; probe++) {
        h = (h + 1) & (C::HC - 1);
        u64 old = atomicCAS(&S.hkey[h], HKEY_EMPTY, key);
        if(old == HKEY_EMPTY) claimed = true;
        if(old == HKEY_EMPTY || old == key) return h;
    }
    return (u32)C::HC;
}

// (the two smallest classes keep the cheap hash for their slab tables of at most 4096 / 8192 entries: the multiplies cost them 2 % of their time)
template <class C> __device__ __forceinline__ u32 early_hash(u64 k) { return (C::EARLY > 8192 ? hash_mix(k) : hash64(k)) & (u32)(C::EARLY - 1); }

// The early-cell table of a slab is never cleared between DP calls: an entry is the word (generation 24 bits | read offset 12 | node 28) -- the level
// is implied by the node -- and only entries of the current generation count; anything else reads as empty.  A DP call that gets its first early cell
// takes the next generation of its slab (dp_iterate).  (Every such call used to clear the whole table first: 32 KB of stores per call in the 16-lane
// class, about half of the kernel's HBM writes on the Graph M workload.)
__device__ __forceinline__ u64 early_word(int gen, u64 key) { return ((u64)(u32)gen << 40) | (key & 0xFFFFFFFFFFull); }

template <class C>
__device__ inline int early_lookup(const DpSlabT<C>& sl, int gen, u64 key)
{
    u32 h = early_hash<C>(key);
    const u64 w = early_word(gen, key);
    for(int probe = 0; probe < C::EARLY; probe++) {
        // entries are published with L2 atomics: read them past the CU's L1
        u64 cur = __hip_atomic_load(&sl.early_key()[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if(cur == w) return __hip_atomic_load(&sl.early_val()[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if((int)(cur >> 40) != gen) return -1;
        h = (h + 1) & (C::EARLY - 1);
    }
    return -1;
}
template <class C>
__device__ inline bool early_insert(const DpSlabT<C>& sl, int gen, u64 key, int slot)
{
    u32 h = early_hash<C>(key);
    const u64 w = early_word(gen, key);
    for(int probe = 0; probe < C::EARLY; probe++) {
        u64 cur = __hip_atomic_load(&sl.early_key()[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        while((int)(cur >> 40) != gen) {              // free (an older generation): claim it; a lane of this group that gets there first leaves a current entry
            const u64 old = atomicCAS(&sl.early_key()[h], cur, w);
            if(old == cur) { cur = w; break; }
            cur = old;
        }
        if(cur == w) { __hip_atomic_store(&sl.early_val()[h], slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return true; }
        h = (h + 1) & (C::EARLY - 1);
    }
    return false;
}

// "x/z" string order of std::set<std::string> achieved_complete_sequence_alignments (extensionAligner.cpp:493, 1431)
// (characters are produced on the fly: no per-lane buffers, no scratch memory)
__device__ inline int dec_len(int v) { int n = 1; while(v >= 10) { v /= 10; n++; } return n; }
__device__ inline int pow10i(int n) { int p = 1; while(n-- > 0) p *= 10; return p; }
// number of decimal digits without a division (the loop of dec_len compiles to one software division per digit)
__device__ __forceinline__ int dec_len_cmp(int v)
{
    return 1 + (v >= 10) + (v >= 100) + (v >= 1000) + (v >= 10000) + (v >= 100000) + (v >= 1000000) + (v >= 10000000) + (v >= 100000000) + (v >= 1000000000);
}
__device__ __forceinline__ u32 pow10_u32(int k)          // 10^k, k = 0 .. 9
{
    u32 p = (k & 1) ? 10u : 1u;
    if(k & 2) p *= 100u;
    if(k & 4) p *= 10000u;
    if(k & 8) p *= 100000000u;
    return p;
}
// order of the decimal strings of a and b, where the character that follows a number ('/' after x, the end of the string after z) sorts before every digit:
// pad the shorter number with zeros to the length of the longer one -- the first differing character decides as the numbers do; equal after padding, the
// shorter one is a proper prefix and comes first.  No division.  (The character-by-character comparison it replaces -- two software divisions per character and
// side -- was a fifth of the 16-lane kernel's instructions in the binary; checked against std::string's operator< on millions of pairs: tools/xz_order_check.py.)
__device__ __forceinline__ int xz_part(int a, int b)       // -1 / 0 / 1
{
    if(a == b) return 0;
    const int la = dec_len_cmp(a), lb = dec_len_cmp(b);
    u64 A = (u64)(u32)a, Bv = (u64)(u32)b;
    if(la < lb) A *= (u64)pow10_u32(lb - la); else if(lb < la) Bv *= (u64)pow10_u32(la - lb);
    if(A != Bv) return A < Bv ? -1 : 1;
    return la < lb ? -1 : 1;
}
// "x1/z1" < "x2/z2" as strings
__device__ __forceinline__ bool xz_less(int x1, int z1, int x2, int z2)
{
    const int c = xz_part(x1, x2);
    if(c) return c < 0;
    return xz_part(z1, z2) < 0;
}

// a number whose unsigned order is the string order of "x/z": symbols end < '/' < '0' .. '9' as digits of a base-12 number of 14 places
// (x < 2^24: 8 digits, '/', z: at most 5 digits -- ranks within a level stay far below 100 000; 12^14 < 2^51)
__device__ inline u64 xz_key(int x, int z)
{
    const int lx = dec_len(x), lz = dec_len(z);
    u64 k = 0; int n = 0;
    for(int i = 0; i < lx; i++, n++) k = k * 12 + (u64)(2 + (x / pow10i(lx - 1 - i)) % 10);
    k = k * 12 + 1; n++;
    for(int i = 0; i < lz && n < 14; i++, n++) k = k * 12 + (u64)(2 + (z / pow10i(lz - 1 - i)) % 10);
    for(; n < 14; n++) k *= 12;
    return k;
}

// one DP item as prepared by k_dp_items (32 bytes)
struct __align__(16) DpItem { int item, rOff, seqLen, start_seq, startLevel, startNode, pad0, pad1; };

// The fused entry point runs the classes from this tier on on its side stream (hlala_api.hip: extend_impl).  Measured on Graph M, 1M pairs per batch
// (tools/gpu_side_ab.sh), two batches in flight / one batch at a time: tier 3: 369 / 406 ms per batch, 4: 360 / 393, 5: 366 / 391, 6: 372 / 379;
// the same two batches without any overlap between them: 400.
// Round 5 (the main stream's kernels take 200 ms of a step, the side stream's 135): with the wide class on the side stream as well, resident step 205.2 -> 199.8 / 200.4 ms (side span 163 ms);
// with the 64-lane class too 211.1 ms (side span 203 ms: the side stream becomes the longer one).  profiles/r05_experiments.txt 10.
#ifndef HLALA_DP_SIDE_TIER
#define HLALA_DP_SIDE_TIER 3
#endif
constexpr int DP_SIDE_TIER = HLALA_DP_SIDE_TIER;
constexpr int DP_LAST_TIER = 6;      // tiers: 0 DpTiny, 1 DpMid, 2 DpSmall, 3 DpWide, 4 DpBroad, 5 DpLarge, 6 DpHuge
constexpr int DP_POOL_TIER = DP_SIDE_TIER > 4 ? DP_SIDE_TIER : 4;      // first class that a tail pool defers and runs once for several batches (k_dp: DpPoolArgs)
constexpr int DP_POOL_MAX = 8;       // batches per pooled launch
// first tier whose class holds a frontier of n cells / a target set of n cells
__device__ __forceinline__ int tier_for_frontier(int n) { return n <= DpMid::WCAP ? 1 : (n <= DpSmall::WCAP ? 2 : (n <= DpWide::WCAP ? 3 : (n <= DpBroad::WCAP ? 4 : (n <= DpLarge::WCAP ? 5 : 6)))); }
__device__ __forceinline__ int tier_for_targets(int n) { return n <= (DpMid::HC * 3) / 4 ? 1 : (n <= (DpSmall::HC * 3) / 4 ? 2 : (n <= (DpWide::HC * 3) / 4 ? 3 : (n <= (DpBroad::HC * 3) / 4 ? 4 : (n <= (DpLarge::HC * 3) / 4 ? 5 : 6)))); }

#define DP_FAIL(code) do { if(gl == 0 && S.err == 0) S.err = (code); } while(0)

template <class C>
__device__ inline int dp_begin(DpLdsT<C>& S, const DpSlabT<C>& sl, const DevGraph& G, const DpItem& it, int itemIdx)
{
    constexpr int GW = C::GW;
    const int gl = grp_lane<GW>();
    // ---- init, :480-519
    for(int i = gl; i < C::HC; i += GW) { S.hkey[i] = HKEY_EMPTY; S.hbest[0][i] = 0; S.hbest[1][i] = 0; S.hbest[2][i] = 0; }
    if(gl == 0) {
        DpState& st = S.st;
        st.itemIdx = itemIdx;
        st.item = it.item; st.rOff = it.rOff; st.seqLen = it.seqLen; st.start_seq = it.start_seq; st.startLevel = it.startLevel; st.startNode = it.startNode;
        st.diagonals = it.seqLen + G.L - 1;
        st.d = 1; st.b1 = 0; st.b2 = 1; st.n1 = 1; st.n2 = 0; st.nCells = 1; st.nCompleted = 0;
        st.curMax = 0; st.firstMaxSlot = 0; st.lastInc = 0; st.earlyInit = 0; st.earlyMaxNat = -1; st.itersRun = 0;
        st.cellsEvaluated = 0;
        st.endSlot = -1; st.endScore = 0; st.nSteps = 0; st.nCols = 0;
        st.have = 0; st.sb = 0; st.se = -1; st.err = 0; st.needTier = 0; st.isAlias = 0;
        S.err = 0; S.nTa = 0; S.jumpMet = 0;
#ifdef HLALA_DP_PROFILE
        S.pfStart = clock64(); S.pfSlow = 0; S.pfImp = 0; S.pfPre = 0; S.pfMaxNT = 0; S.pfMaxF = 0; for(int i = 0; i < 16; i++) S.pfPh[i] = 0;
#endif
        CellRec c0; c0.key = mk_key(it.startLevel, it.start_seq, it.startNode);
        c0.sc[0] = 0; c0.sc[1] = (short)DP_NEG; c0.sc[2] = (short)DP_NEG; c0.sc[3] = 0; c0.bt[0] = 0; c0.bt[1] = 0; c0.bt[2] = 0; c0.pad = 0;
        sl.cell()[0] = c0;
        S.fkey[0][0] = mk_key(it.startLevel, it.start_seq, it.startNode); S.fslot[0][0] = 0;
        S.fD[0][0] = 0; S.fG[0][0] = (short)DP_NEG; S.fS[0][0] = (short)DP_NEG;
    }
    DSYNC();
    return PH_RUN;
}

// One iteration of extensionAligner::fullNeedleman_diagonal_extension_gapJumper (extensionAligner.cpp:531-1105) with
// returnGlobalScore = false, preferSequenceCompleAlignments = true, empty blockedPathsTable,
// diagonal_stop_threshold = -16 (the only configuration extendSeedChain uses, :229-241, :281-293).
template <class C>
__device__ inline int dp_iterate(DpLdsT<C>& S, const DpSlabT<C>& sl, const DevGraph& G, const int4* nrec, const uint8_t* readBases, const bool fwd, int& edgesAcc, u64* sortScratch)
{
    constexpr int GW = C::GW;
    const int gl = grp_lane<GW>();
    DpState& st = S.st;
    const int dir = fwd ? 1 : -1;
    const int seqLen = guni<GW>(st.seqLen);
    const int max_levelI = G.L - 1, max_seqI = seqLen;       // :431-463 (min_* are 0 in both directions)
    const int limitY = fwd ? seqLen : 0;
    const int d = guni<GW>(st.d);
    const int n1 = guni<GW>(st.n1), n2 = guni<GW>(st.n2);
    const int b1 = guni<GW>(st.b1), b2 = guni<GW>(st.b2), bn = b2;       // (bn: where the new frontier goes)
    const int lastInc0 = guni<GW>(st.lastInc), diagonals = guni<GW>(st.diagonals);
    const uint8_t* seqp = readBases + guni<GW>(st.rOff);

    // ---- loop header, :531-560
    if(d > diagonals || (d - lastInc0) > 40) return PH_SELECT;                             // :553 maximum_steps_nonIncrease
    if(n1 == 0 && n2 == 0) {
        // both frontiers empty: the remaining iterations of the reference loop are no-ops
        int last = lastInc0 + 40; if(last > diagonals) last = diagonals;
        if(gl == 0) st.itersRun = last;
        DSYNC();
        return PH_SELECT;
    }
    if(d > 60000) { if(gl == 0) { st.itersRun = d; st.err = __LINE__; } DSYNC(); return PH_DONE; }      // watchdog: far beyond any read length + patience

#if defined(HLALA_DP_TIMING)
    long long tq0 = clock64();
#define DP_TQ(i) do { __builtin_amdgcn_s_waitcnt(0); long long t_ = clock64(); if(gl == 0) S.tPh[i] += t_ - tq0; tq0 = t_; } while(0)
#elif defined(HLALA_DP_PROFILE)
    long long tq0 = clock64();      // phases of the profiled DP: 4 records, 5 pushes, 0 target list, 3 early look-ups, 6 evaluate passes, 1 post-evaluate, 2 filter + sort + reset
#define DP_TQ(i) do { __builtin_amdgcn_s_waitcnt(0); long long t_ = clock64(); if(gl == 0) S.pfPh[i] += t_ - tq0; tq0 = t_; } while(0)
#else
#define DP_TQ(i) do { } while(0)
#endif
    // ================= generate =====================================================
    // Lane i serves entry i of the m-2 frontier (match / mismatch, :565-607) and entry i of the m-1 frontier (gaps and jumps,
    // :613-787).  Each needs ONE 32-byte node record (first CSR edge, degree, jumps, the first two edges, the first jump:
    // flat_graph.hpp); nodes with more than two edges or more than one jump read the rest from the CSR arrays.  The push
    // index of a candidate fixes its precedence (first maximum in push order), so the order in which lanes and loops
    // execute the pushes is irrelevant.
    const int* eoff = fwd ? G.out_off : G.in_off; const int* eto = fwd ? G.out_to : G.in_from; const uint8_t* elab = fwd ? G.out_label : G.in_label;
    const int* joff = fwd ? G.jf_off : G.jb_off; const int* jnode = fwd ? G.jf_node : G.jb_node; const int* jlvl = fwd ? G.jf_lvl : G.jb_lvl;
    // The low bits of a push index hold the RANK of the edge (jump) among the node's earlier edges (jumps) to the same target node, not the edge number:
    // candidates of different edges of one frontier cell compete only when they reach the same cell, i.e. the same node, and among those the rank orders
    // exactly as the edge number does.  A node may therefore have any number of edges and gap-path jumps (allele-rich levels of a real PRG: hundreds); the
    // field limits only the parallel edges between ONE pair of nodes (DP_MAX_PARALLEL, checked in hlala_create).  The backtrace resolves (previous node,
    // this node, rank) to the edge.
    const uint8_t* eprk = fwd ? G.out_prank : G.in_prank; const uint8_t* jprk = fwd ? G.jf_prank : G.jb_prank;
    int edges = 0;      // (the direction is wave-uniform: these are scalar selects)
    const int nMax = n1 > n2 ? n1 : n2;
    // One wave per DP and a table of 512+ entries: the list of this iteration's targets is written as the entries are claimed (wave ballot, running count
    // in a register) instead of being collected from the table afterwards -- 8 .. 256 rounds over mostly empty entries.  Its order is arbitrary either way.
    constexpr bool APPEND = (GW >= 64) && (C::WCAP > 64);
    typedef typename TlistT<(C::HC <= 256)>::type TlT;
    int nTa = 0;
    auto append = [&](const bool claimed, const u32 h) {
        if constexpr (APPEND && GW == 64) {
            const u64 m = __ballot(claimed);
            if(claimed) S.tlist[nTa + (int)__builtin_amdgcn_mbcnt_hi((u32)(m >> 32), __builtin_amdgcn_mbcnt_lo((u32)m, 0u))] = (TlT)h;
            nTa += __popcll(m);
        } else if constexpr (APPEND) {
            // several waves append to one list: a wave reserves its entries with one atomic on the shared count (the order of the list is arbitrary)
            const u64 m = __ballot(claimed);
            if(m) {
                int base = 0;
                if(lane_id() == 0) base = atomicAdd(&S.nTa, (int)__popcll(m));
                base = __builtin_amdgcn_readfirstlane(base);
                const int pos = base + (int)__builtin_amdgcn_mbcnt_hi((u32)(m >> 32), __builtin_amdgcn_mbcnt_lo((u32)m, 0u));
                if(claimed && pos < C::HC) S.tlist[pos] = (TlT)h;
            }
        }
    };
    for(int i = gl; i < nMax; i += GW) {
        // ---- round 0: frontier entries (LDS)
        const bool hasA = i < n2, hasB = i < n1;
        u64 pkA = hasA ? S.fkey[b2][i] : 0, pkB = hasB ? S.fkey[b1][i] : 0;
        const int pxA = key_x(pkA), pyA = key_y(pkA), nodeA = key_node(pkA);
        const int pxB = key_x(pkB), pyB = key_y(pkB), nodeB = key_node(pkB);
        const int nxA = pxA + dir, nyA = pyA + dir;
        const bool doA = hasA && !(nxA > max_levelI || nyA > max_seqI || nxA < 0 || nyA < 0);
        const int pDA = hasA ? (int)S.fD[b2][i] : 0;
        const int pD = hasB ? (int)S.fD[b1][i] : 0, pG = hasB ? (int)S.fG[b1][i] : 0, pS = hasB ? (int)S.fS[b1][i] : 0;
        // ---- round 1: node records, read character
        int4 ra0 = make_int4(0, 0, 0, 0), ra1 = ra0, rb0 = ra0, rb1 = ra0; unsigned char rc = 0;
        if(doA) { ra0 = nrec[2 * (size_t)nodeA]; ra1 = nrec[2 * (size_t)nodeA + 1]; rc = fwd ? seqp[pyA] : seqp[pyA - 1]; }
        if(hasB) { rb0 = nrec[2 * (size_t)nodeB]; rb1 = nrec[2 * (size_t)nodeB + 1]; }
        const int a0 = ra0.x, degA = ra0.y & 0xFFFF, tnA0 = ra0.z, tnA1 = ra0.w;
        const unsigned char labA0 = (unsigned char)(ra1.w & 0xFF), labA1 = (unsigned char)((ra1.w >> 8) & 0xFF);
        const int rkA1 = (ra1.w >> 16) & 1, rkB1 = (rb1.w >> 16) & 1;          // rank of edge 1: 1 iff it leads where edge 0 leads
        const int e0 = rb0.x, degB = rb0.y & 0xFFFF, tnB0 = rb0.z, tnB1 = rb0.w;
        const int j0 = rb1.x, j1 = j0 + (int)((u32)rb0.y >> 16), jn0 = rb1.y, jx0 = rb1.z;
        const unsigned char labB0 = (unsigned char)(rb1.w & 0xFF), labB1 = (unsigned char)((rb1.w >> 8) & 0xFF);
        DP_TQ(4);
        // ---- six target cells per lane, claimed with six independent compare-and-swaps issued back to back (one LDS round
        // trip for all), then the values: 0/1 = edges 0/1 of the m-2 entry (D); 2 = gap in graph (:621-661, both GG candidates);
        // 3/4 = edges 0/1 of the m-1 entry (gap in sequence :664-754: both SG candidates, or SG + the non-affine D across a
        // '_' edge, :738-752); 5 = first gap-path jump (:757-786, jump_length * S_graphGap = 0)
        const int ord0 = (1 << (C::IBITS + 8)) | (i << 8);
        const int nyG = pyB + dir, nxB = pxB + dir;
        const bool okA = doA, okB = hasB;
        const bool sgB = okB && nxB >= 0 && nxB <= max_levelI;
        typedef typename C::Best BestT;
        {   // batch 1: both edges of the m-2 entry, the graph gap
            bool cv[3]; u64 ck[3]; u32 ch[3]; u64 cold[3];
            cv[0] = okA && degA > 0; ck[0] = mk_key(nxA, nyA, tnA0);
            cv[1] = okA && degA > 1; ck[1] = mk_key(nxA, nyA, tnA1);
            cv[2] = hasB && nyG >= 0 && nyG <= max_seqI; ck[2] = mk_key(pxB, nyG, nodeB);
#pragma unroll
            for(int q = 0; q < 3; q++) { ch[q] = tgt_hash<C>(ck[q]); cold[q] = HKEY_EMPTY; if(cv[q]) cold[q] = atomicCAS(&S.hkey[ch[q]], HKEY_EMPTY, ck[q]); }
            if constexpr (APPEND) {
                bool cl[3];
#pragma unroll
                for(int q = 0; q < 3; q++) {
                    cl[q] = cv[q] && cold[q] == HKEY_EMPTY;
                    if(cv[q] && !(cold[q] == HKEY_EMPTY || cold[q] == ck[q])) { ch[q] = dp_probe<C>(S, ck[q], ch[q], cl[q]); if(ch[q] >= (u32)C::HC) { S.err = __LINE__; cv[q] = false; cl[q] = false; } }
                }
#pragma unroll
                for(int q = 0; q < 3; q++) append(cl[q], ch[q]);
            } else {
#pragma unroll
                for(int q = 0; q < 3; q++)
                    if(cv[q] && !(cold[q] == HKEY_EMPTY || cold[q] == ck[q])) { ch[q] = dp_probe<C>(S, ck[q], ch[q]); if(ch[q] >= (u32)C::HC) { S.err = __LINE__; cv[q] = false; } }
            }
            if(cv[0]) { BestT v; pack_best<C>(v, pDA + (labA0 == rc ? 2 : -5), (i << 8) | 0); atomicMax(&S.hbest[M_D][ch[0]], v); }
            if(cv[1]) { BestT v; pack_best<C>(v, pDA + (labA1 == rc ? 2 : -5), (i << 8) | rkA1); atomicMax(&S.hbest[M_D][ch[1]], v); }
            if(cv[2]) { BestT v, w; pack_best<C>(v, pD - 6, ord0 | 0); if(pG != DP_NEG) { pack_best<C>(w, pG - 2, ord0 | 1); if(w > v) v = w; } atomicMax(&S.hbest[M_GG][ch[2]], v); }
        }
        {   // batch 2: both edges of the m-1 entry, the first jump
            bool cv[3]; u64 ck[3]; u32 ch[3]; u64 cold[3];
            cv[0] = sgB && degB > 0; ck[0] = mk_key(nxB, pyB, tnB0);
            cv[1] = sgB && degB > 1; ck[1] = mk_key(nxB, pyB, tnB1);
            if constexpr (C::JF) { cv[2] = false; ck[2] = 0; if(okB && j1 > j0) { S.err = __LINE__; S.jumpMet = 1; } }       // a jump after all: the call is re-run in the general instantiation
            else { cv[2] = okB && j1 > j0 && jx0 >= 0 && jx0 <= max_levelI; ck[2] = mk_key(jx0, pyB, jn0); }
#pragma unroll
            for(int q = 0; q < 3; q++) { ch[q] = tgt_hash<C>(ck[q]); cold[q] = HKEY_EMPTY; if(cv[q]) cold[q] = atomicCAS(&S.hkey[ch[q]], HKEY_EMPTY, ck[q]); }
            if constexpr (APPEND) {
                bool cl[3];
#pragma unroll
                for(int q = 0; q < 3; q++) {
                    cl[q] = cv[q] && cold[q] == HKEY_EMPTY;
                    if(cv[q] && !(cold[q] == HKEY_EMPTY || cold[q] == ck[q])) { ch[q] = dp_probe<C>(S, ck[q], ch[q], cl[q]); if(ch[q] >= (u32)C::HC) { S.err = __LINE__; cv[q] = false; cl[q] = false; } }
                }
#pragma unroll
                for(int q = 0; q < 3; q++) append(cl[q], ch[q]);
            } else {
#pragma unroll
                for(int q = 0; q < 3; q++)
                    if(cv[q] && !(cold[q] == HKEY_EMPTY || cold[q] == ck[q])) { ch[q] = dp_probe<C>(S, ck[q], ch[q]); if(ch[q] >= (u32)C::HC) { S.err = __LINE__; cv[q] = false; } }
            }
#pragma unroll
            for(int kk = 0; kk < 2; kk++) {
                const unsigned char lab = kk ? labB1 : labB0;
                const int rk = kk ? rkB1 : 0;
                if(cv[kk]) {
                    BestT v, w;
                    if(lab != '_') {
                        pack_best<C>(v, pD - 6, ord0 | (2 * rk)); if(pS != DP_NEG) { pack_best<C>(w, pS - 2, ord0 | (2 * rk + 1)); if(w > v) v = w; }
                        atomicMax(&S.hbest[M_SG][ch[kk]], v);
                    } else {
                        if(pS != DP_NEG) { pack_best<C>(w, pS, ord0 | (2 * rk + 1)); atomicMax(&S.hbest[M_SG][ch[kk]], w); }
                        pack_best<C>(v, pD, ord0 | rk); atomicMax(&S.hbest[M_D][ch[kk]], v);
                    }
                }
            }
            if constexpr (!C::JF) if(cv[2]) { BestT v; pack_best<C>(v, pD, ord0 | 128); atomicMax(&S.hbest[M_D][ch[2]], v); }
        }
        if(okA) edges += degA;
        if(sgB) edges += degB;
        // ---- the rest of wide nodes: edges 2.. and jumps 1.., read from the CSR arrays.  (With the claim-time target list the loops run to the
        // largest count among the lanes, so that every lane takes part in the ballots.)
        if constexpr (APPEND) {
            for(int k = 2; __ballot(okA && k < degA) != 0; k++) {
                int at = -1;
                if(okA && k < degA) {
                    int tn = eto[a0 + k]; unsigned char lab = elab[a0 + k];
                    if(!dp_push<C>(S, mk_key(nxA, nyA, tn), M_D, pDA + (lab == rc ? 2 : -5), (i << 8) | (int)eprk[a0 + k], at)) S.err = __LINE__;
                }
                append(at >= 0, (u32)at);
            }
            for(int kk = 2; __ballot(sgB && kk < degB) != 0; kk++) {
                int at = -1;
                if(sgB && kk < degB) {
                    int tn = eto[e0 + kk]; unsigned char lab = elab[e0 + kk]; const int rk = (int)eprk[e0 + kk];
                    u64 k = mk_key(nxB, pyB, tn);
                    if(lab != '_') {
                        if(!dp_push<C>(S, k, M_SG, pD - 6, ord0 | (2 * rk), at)) S.err = __LINE__;
                        if(pS != DP_NEG) if(!dp_push<C>(S, k, M_SG, pS - 2, ord0 | (2 * rk + 1), at)) S.err = __LINE__;
                    } else {
                        if(pS != DP_NEG) if(!dp_push<C>(S, k, M_SG, pS, ord0 | (2 * rk + 1), at)) S.err = __LINE__;
                        if(!dp_push<C>(S, k, M_D, pD, ord0 | rk, at)) S.err = __LINE__;
                    }
                }
                append(at >= 0, (u32)at);
            }
            for(int j = j0 + 1; __ballot(okB && j < j1) != 0; j++) {
                int at = -1;
                if(okB && j < j1) {
                    int tn = jnode[j]; int jx = jlvl[j];
                    if(!(jx < 0 || jx > max_levelI)) if(!dp_push<C>(S, mk_key(jx, pyB, tn), M_D, pD, ord0 | (128 + (int)jprk[j]), at)) S.err = __LINE__;
                }
                append(at >= 0, (u32)at);
            }
        } else {
            if(okA) for(int k = 2; k < degA; k++) {
                int tn = eto[a0 + k]; unsigned char lab = elab[a0 + k];
                if(!dp_push<C>(S, mk_key(nxA, nyA, tn), M_D, pDA + (lab == rc ? 2 : -5), (i << 8) | (int)eprk[a0 + k])) S.err = __LINE__;
            }
            if(sgB) for(int kk = 2; kk < degB; kk++) {
                int tn = eto[e0 + kk]; unsigned char lab = elab[e0 + kk]; const int rk = (int)eprk[e0 + kk];
                u64 k = mk_key(nxB, pyB, tn);
                if(lab != '_') {
                    if(!dp_push<C>(S, k, M_SG, pD - 6, ord0 | (2 * rk))) S.err = __LINE__;
                    if(pS != DP_NEG) if(!dp_push<C>(S, k, M_SG, pS - 2, ord0 | (2 * rk + 1))) S.err = __LINE__;
                } else {
                    if(pS != DP_NEG) if(!dp_push<C>(S, k, M_SG, pS, ord0 | (2 * rk + 1))) S.err = __LINE__;
                    if(!dp_push<C>(S, k, M_D, pD, ord0 | rk)) S.err = __LINE__;
                }
            }
            if constexpr (!C::JF) if(okB) for(int j = j0 + 1; j < j1; j++) {
                int tn = jnode[j]; int jx = jlvl[j];
                if(jx < 0 || jx > max_levelI) continue;
                if(!dp_push<C>(S, mk_key(jx, pyB, tn), M_D, pD, ord0 | (128 + (int)jprk[j]))) S.err = __LINE__;
            }
        }
    }
    edgesAcc += edges;
    DSYNC();
    DP_TQ(5);
    // target list = occupied hash entries, compacted with a ballot per GW entries (no per-push counter, no ordering assumed)
    int nT = 0;
    if constexpr (APPEND && GW == 64) nT = __builtin_amdgcn_readfirstlane(nTa);       // (lane 0 has been through every round of the loop above)
    else if constexpr (APPEND) { nT = guni<GW>(S.nTa); if(nT > C::HC) nT = C::HC + 1; }
    else if constexpr (C::IN_MEMORY) {
        // (the table is in HBM: four independent loads per lane and trip instead of one)
        for(int h0 = 0; h0 < C::HC; h0 += 4 * GW) {
            u64 kk[4];
#pragma unroll
            for(int q = 0; q < 4; q++) kk[q] = __hip_atomic_load(&S.hkey[h0 + q * GW + gl], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
            for(int q = 0; q < 4; q++) {
                const bool occ = kk[q] != HKEY_EMPTY;
                int tot; const int rk = grp_rank<GW>(occ, tot);
                if(occ) S.tlist[nT + rk] = (typename TlistT<(C::HC <= 256)>::type)(h0 + q * GW + gl);
                nT += tot;
            }
        }
    } else
    for(int h0 = 0; h0 < C::HC; h0 += GW) {
        const int h = h0 + gl;
        const bool occ = S.hkey[h] != HKEY_EMPTY;
        int tot; const int rk = grp_rank<GW>(occ, tot);
        if(occ) S.tlist[nT + rk] = (typename TlistT<(C::HC <= 256)>::type)h;
        nT += tot;
    }
    DSYNC();
    DP_TQ(0);
    if(nT > (C::HC * 3) / 4 || guni<GW>(S.err)) { if(gl == 0) { st.itersRun = d; st.err = __LINE__; if(nT > (C::HC * 3) / 4) st.needTier = tier_for_targets(nT); } DSYNC(); return PH_DONE; }

    // ================= evaluate =====================================================
    // (the part of the DP state that only this phase needs is read here, not at the top: shorter live ranges)
    const int startLevel = guni<GW>(st.startLevel), start_seq = guni<GW>(st.start_seq);
    const int curMax0 = guni<GW>(st.curMax);
    const int nCompleted0 = guni<GW>(st.nCompleted);
    int nCells = guni<GW>(st.nCells);
    int earlyInit = guni<GW>(st.earlyInit);
    const int earlyMaxNat0 = guni<GW>(st.earlyMaxNat);
    if(gl == 0) { S.nNew = 0; S.nImp = 0; S.nCompletedAdd = 0; S.nTa = 0; }       // (nTa: read above, counts again from the next generate phase on)
    DSYNC();
    int itMaxNew = DP_NEG;        // max Dv over kept targets of this iteration
    u64 itMaxKey = ~0ull;         // smallest key achieving it (= first such cell in std::map order)
    bool anyEqDiff = false, anyOw = false, anyExisting = false;

    // Cells reached through a gap-path jump arrive EARLIER than their natural diagonal |dx|+|dy| and can be reached again later ("scores"
    // merge, :951-979): they are registered with the DP's cell hash in the slab (HBM).  Every target of iteration d has a natural diagonal
    // >= d -- a cell created on its natural diagonal is never met again -- so a target can only meet an early cell while d <= the largest
    // natural diagonal of the early cells created so far: in those iterations every kept target is looked up (a probing read), and only cells
    // that are early themselves are inserted.  tes[t] = table slot of an existing cell, or -1.  (Tried: a Bloom filter in LDS in front of
    // the look-ups -- no gain on Graph M, a loss on gap-heavy small graphs whose filter saturates.)
    const bool prepass = !C::JF && earlyInit && d <= earlyMaxNat0;
    int earlyNatMax = -1;         // per lane: largest natural diagonal of the early cells this iteration creates
    if(prepass) {
        for(int t0 = 0; t0 < nT; t0 += GW) {
            int t = t0 + gl; int es = -1;
            if(t < nT) {
                int h = S.tlist[t];
                int Dv = max(best_score<C>(S.hbest[M_D][h]), max(best_score<C>(S.hbest[M_GG][h]), best_score<C>(S.hbest[M_SG][h])));
                if(Dv >= -16) es = early_lookup<C>(sl, earlyInit, S.hkey[h]);
                S.tes[t] = (typename C::Slot)es;
            }
            if(es >= 0) anyExisting = true;
        }
        anyExisting = grp_any<GW>(anyExisting);
        DSYNC();
    }
    DP_TQ(3);
    const bool slow = anyExisting;
    const bool hadEarly = prepass;             // S.tes[] holds lookups only if the pre-pass ran
    bool failed = false;

    for(int pass = 0; pass < (slow ? 2 : 1); pass++) {
        for(int t0 = 0; t0 < nT; t0 += GW) {
            int t = t0 + gl;
            bool act = t < nT;
            int h = act ? S.tlist[t] : 0;
            u64 key = act ? S.hkey[h] : 0;
            typename C::Best bD = act ? S.hbest[M_D][h] : 0, bG = act ? S.hbest[M_GG][h] : 0, bS = act ? S.hbest[M_SG][h] : 0;
            int Dc = best_score<C>(bD), GGv = best_score<C>(bG), SGv = best_score<C>(bS);
            int Dv = Dc, dsel = 0;                      // D candidates first, then GG, then SG (:840-865); first maximum wins
            if(GGv > Dv) { Dv = GGv; dsel = 1; }
            if(SGv > Dv) { Dv = SGv; dsel = 2; }
            bool keep = act && (Dv >= -16);                                               // :949
            DP_TQ(8);
            int es = -1; bool isNew; int slot;
            if(pass == 0) {
                if(hadEarly && act) S.hq[h] = (typename C::ImpIdx)~0u;
                if(hadEarly && keep) { int v = S.tes[t]; if(v >= 0) es = v; }
                isNew = keep && es < 0;
                int total; int off = grp_excl_scan<GW>(isNew ? 1 : 0, total);
                slot = isNew ? nCells + off : es;
                if(nCells + total > C::CELLS) { DP_FAIL(__LINE__); failed = true; }
                nCells += total;
            } else {
                slot = keep ? S.tes[t] : -1;
                isNew = keep && (S.timp[t] & 0x80);
                es = isNew ? -1 : slot;
            }
            const bool ok = !failed && S.err == 0;
            DP_TQ(9);
            // ---- back pointers of the three matrices, decoded from the winning push index
            u32 btD = 0, btG = 0, btS = 0;
            int srcScore = 0;       // score the real previous step came from (fast form of the `diff` rule)
            if(keep && ok && slot >= 0 && slot < C::CELLS) {
                // back pointer = (previous cell slot, source matrix, kind, local push index j); the graph edge / gap path behind j
                // is resolved only for the cells on the final path, at backtrace time
                if(bG) { int o = best_order<C>(bG); int i = (o >> 8) & ((1 << C::IBITS) - 1); int j = o & 255;
                         btG = mk_bt(S.fslot[b1][i], j ? 1 : 0, K_GGAP, -1); }
                if(bS) { int o = best_order<C>(bS); int i = (o >> 8) & ((1 << C::IBITS) - 1); int j = o & 255;
                         btS = mk_bt(S.fslot[b1][i], (j & 1) ? 2 : 0, K_SGAP, j >> 1); }
                if(dsel == 0) {
                    int o = best_order<C>(bD); int ph = o >> (C::IBITS + 8); int i = (o >> 8) & ((1 << C::IBITS) - 1); int j = o & 255;
                    int sb = ph ? b1 : b2;
                    srcScore = S.fD[sb][i];
                    if(!ph) btD = mk_bt(S.fslot[sb][i], 0, K_DIAG, j);
                    else if(j < 128) btD = mk_bt(S.fslot[sb][i], 0, K_SGAP, j);
                    else btD = mk_bt(S.fslot[sb][i], 0, K_JUMP, j - 128);
                } else if(dsel == 1) {
                    btD = mk_bt(slot, 1, K_HOP, -1);
                    int o = best_order<C>(bG); int i = (o >> 8) & ((1 << C::IBITS) - 1); int j = o & 255;
                    srcScore = j ? S.fG[b1][i] : S.fD[b1][i];
                } else {
                    btD = mk_bt(slot, 2, K_HOP, -1);
                    int o = best_order<C>(bS); int i = (o >> 8) & ((1 << C::IBITS) - 1); int j = o & 255;
                    srcScore = (j & 1) ? S.fS[b1][i] : S.fD[b1][i];
                }
            }
            DP_TQ(10);
            int impMask = 0;
            int mD = Dv, mG = GGv, mS = SGv;          // merged values
            u32 mbtD = btD;                            // merged D back pointer
            if(pass == 0) {
                // ---- new cells: write the table entry, register early / sequence-complete cells
                if(isNew && ok && slot < C::CELLS) {
                    uint4* dst = (uint4*)(sl.cell() + slot);
                    dst[0] = make_uint4((u32)key, (u32)(key >> 32), ((u32)(unsigned short)(short)Dv) | ((u32)(unsigned short)(short)GGv << 16), (u32)(unsigned short)(short)SGv);
                    dst[1] = make_uint4(btD, btG, btS, 0u);
                }
                int x = key_x(key), y = key_y(key);
                int natural = (x > startLevel ? x - startLevel : startLevel - x) + (y > start_seq ? y - start_seq : start_seq - y);
                bool isEarly = !C::JF && isNew && natural > d;
                if(isEarly && natural > earlyNatMax) earlyNatMax = natural;
                if(grp_any<GW>(isEarly)) {                       // early cells: into the slab hash
                    if(!earlyInit) {
                        // first early cell of this call: the next generation of the slab's table (early_lookup)
                        int g = 0;
                        if(gl == 0) g = __hip_atomic_load(sl.early_gen(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1;
                        g = grp_bcast0<GW>(g);
                        if(g > DP_EARLY_GEN_MAX) { for(int i = gl; i < C::EARLY; i += GW) sl.early_key()[i] = 0; g = 1; }
                        if(gl == 0) __hip_atomic_store(sl.early_gen(), g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        earlyInit = g;
                        DSYNC();
                    }
                    if(isEarly) if(!early_insert<C>(sl, earlyInit, key, slot)) S.err = __LINE__;
                }
                if(isNew && y == limitY) {                                                     // :982-999
                    int pos = atomicAdd(&S.nCompletedAdd, 1);
                    if(nCompleted0 + pos < C::COMPLETED) sl.completed()[nCompleted0 + pos] = slot; else S.err = __LINE__;
                }
            }
            DP_TQ(11);
            // ---- existing cells: each matrix independently overwritten iff strictly greater, :951-979 (writes are staged)
            if constexpr (!C::JF) if(keep && !isNew && ok) {
                const CellRec* er = sl.cell() + es;
                int oD = er->sc[0], oG = er->sc[1], oS = er->sc[2];
                if(Dv > oD) impMask |= 1; else { mD = oD; mbtD = er->bt[0]; }
                if(GGv > oG) impMask |= 2; else mG = oG;
                if(SGv > oS) impMask |= 4; else mS = oS;
                if(impMask && pass == 0) {
                    int p = atomicAdd(&S.nImp, 1);
                    if(p < C::IMPCAP) {
                        S.hq[h] = (typename C::ImpIdx)p;
                        sl.imp_slot()[p] = es; sl.imp_key()[p] = key; sl.imp_mask()[p] = impMask;
                        sl.imp_new()[4 * p + 0] = (short)mD; sl.imp_new()[4 * p + 1] = (short)mG; sl.imp_new()[4 * p + 2] = (short)mS;
                        sl.imp_bt()[3 * p + 0] = btD; sl.imp_bt()[3 * p + 1] = btG; sl.imp_bt()[3 * p + 2] = btS;
                    } else S.err = __LINE__;
                }
            }
            if(pass == 0 && slow) {
                // exact diff needs every staged improvement of the iteration: finish in the second pass
                if(act) { S.tes[t] = (typename C::Slot)slot; S.timp[t] = (unsigned char)(impMask | (isNew ? 0x80 : 0)); }
                continue;
            }
            DP_TQ(12);
            if(impMask != 0) anyOw = true;            // (per lane; combined over the group after the loop)
            // ---- the `diff` rule, :1007-1041: score difference to the real previous step of the MERGED D back pointer
            int diff = 1;
            if(keep && ok) {
                if(!slow) {
                    diff = Dv - srcScore;       // no table entry changes this iteration: cached frontier values are the table values
                } else {
                    u32 b = mbtD;
                    int guard = 0;
                    while(bt_kind(b) == K_HOP && guard++ < 4) {
                        int m = bt_src(b);
                        bool useNew = isNew || (impMask & (1 << m));
                        if(useNew) b = (m == 1) ? btG : btS; else b = sl.cell()[slot].bt[m];
                    }
                    int ps = bt_prev(b), pm = bt_src(b);
                    int pv = sl.cell()[ps].sc[pm];
                    // a predecessor improved in THIS iteration counts with its new value only if it precedes this cell in map order
                    // (the improved cells of an iteration are targets of it: found through the target hash, not by scanning the staged list)
                    const u64 pkey = sl.cell()[ps].key;
                    if(pkey < key) {
                        const int ph = dp_find<C>(S, pkey);
                        if(ph >= 0) { const int q = (int)S.hq[ph]; if(q < C::IMPCAP && q < S.nImp && (sl.imp_mask()[q] & (1 << pm))) pv = sl.imp_new()[4 * q + pm]; }
                    }
                    diff = Dv - pv;
                }
            }
            // ---- running maximum bookkeeping, :1043-1062
            if(keep && Dv == curMax0 && diff != 0) anyEqDiff = true;
            // per lane: the largest D among its kept targets and the smallest key that carries it (combined over the group after the loop)
            if(keep) { if(Dv > itMaxNew) { itMaxNew = Dv; itMaxKey = key; } else if(Dv == itMaxNew && key < itMaxKey) itMaxKey = key; }
            // stash for the filter phase: [0] = slot (or ~0 if dropped), [1] = merged D | GG<<16, [2] = merged SG
            if(act) {
                S.hbest[0][h] = keep ? (typename C::Best)(u32)slot : (typename C::Best)0xFFFFFFFFu;
                S.hbest[1][h] = (typename C::Best)(((u32)(unsigned short)(short)mD) | ((u32)(unsigned short)(short)mG << 16));
                S.hbest[2][h] = (typename C::Best)(u32)(unsigned short)(short)mS;
            }
            DP_TQ(13);
        }
        DSYNC();
    }
    // the group's view of what its lanes saw: a kept target equal to the running maximum with a real step behind it, an overwritten entry, the iteration's
    // maximum and -- only needed when the maximum moves -- the first cell in map order that carries it
    anyEqDiff = grp_any<GW>(anyEqDiff); anyOw = grp_any<GW>(anyOw);
    {
        const int mine = itMaxNew;
        itMaxNew = grp_max_i32<GW>(mine);
        if(itMaxNew > curMax0) itMaxKey = grp_min_u64<GW>(mine == itMaxNew ? itMaxKey : ~0ull);
    }
    DP_TQ(6);
    if(guni<GW>(S.err)) { if(gl == 0) { st.itersRun = d; st.err = __LINE__; } DSYNC(); return PH_DONE; }
    const int nCompletedNew = nCompleted0 + guni<GW>(S.nCompletedAdd);
    // apply staged improvements of existing cells and patch cached frontier copies
    if constexpr (!C::JF) {
        int nImp = guni<GW>(S.nImp);
        for(int q = gl; q < nImp; q += GW) {
            int es = sl.imp_slot()[q]; int msk = sl.imp_mask()[q];
            for(int m = 0; m < 3; m++) if(msk & (1 << m)) { sl.cell()[es].sc[m] = sl.imp_new()[4 * q + m]; sl.cell()[es].bt[m] = sl.imp_bt()[3 * q + m]; }
        }
        if(nImp) {
            DSYNC();
            // cached copies of the improved cells in the frontier that lives on (b1; b2 is about to be overwritten): a frontier cell that was improved is
            // a target of this iteration
            for(int i = gl; i < n1; i += GW) {
                const int ph = dp_find<C>(S, S.fkey[b1][i]);
                if(ph >= 0) { const int q = (int)S.hq[ph]; if(q < C::IMPCAP && q < nImp) { S.fD[b1][i] = sl.imp_new()[4 * q + 0]; S.fG[b1][i] = sl.imp_new()[4 * q + 1]; S.fS[b1][i] = sl.imp_new()[4 * q + 2]; } }
            }
        }
    }
    // "== currentMaximum && diff != 0" / "> currentMaximum" / overwritten entry all set lastMaximumIncrease_at_diagonalI
    int curMax = curMax0, lastInc = lastInc0, firstMaxSlot = -1;
    if(itMaxNew > curMax0) {
        curMax = itMaxNew; lastInc = d;
        int fs = -1;      // slot of the first cell in map order that carries the new maximum
        for(int t = gl; t < nT; t += GW) { int h = S.tlist[t]; if(S.hkey[h] == itMaxKey) fs = (int)S.hbest[0][h]; }
        firstMaxSlot = grp_max_i32<GW>(fs);
    }
    if(anyEqDiff || anyOw) lastInc = d;
    int earlyMaxNat = earlyMaxNat0;
    if constexpr (!C::JF) if(earlyInit) { const int m = grp_max_i32<GW>(earlyNatMax); if(m > earlyMaxNat) earlyMaxNat = m; }

    DP_TQ(1);
    // ================= filter + sort, :1076-1105 ======================================
    int mx = itMaxNew;          // without merges the merged D of a kept target is its new D
    if constexpr (!C::JF) if(slow) {
        mx = DP_NEG;
        for(int t = gl; t < nT; t += GW) { int h = S.tlist[t]; if((u32)S.hbest[0][h] != 0xFFFFFFFFu) { int v = (short)((u32)S.hbest[1][h] & 0xFFFF); mx = max(mx, v); } }
        mx = grp_max_i32<GW>(mx);
    }
    int nNew = 0;
    // survivors are first compacted into the new frontier buffer in target-list order ...
    for(int t0 = 0; t0 < nT; t0 += GW) {
        int t = t0 + gl;
        bool pass = false; u64 key = 0; int h = 0;
        if(t < nT) { h = S.tlist[t]; key = S.hkey[h]; if((u32)S.hbest[0][h] != 0xFFFFFFFFu) { int v = (short)((u32)S.hbest[1][h] & 0xFFFF); pass = (mx - v) <= 15; } }
        int passTot; const int pos = nNew + grp_rank<GW>(pass, passTot);
        if(pass && pos < C::WCAP) {
            S.fkey[bn][pos] = key; S.fslot[bn][pos] = (typename C::Slot)(int)S.hbest[0][h];
            S.fD[bn][pos] = (short)((u32)S.hbest[1][h] & 0xFFFF); S.fG[bn][pos] = (short)((u32)S.hbest[1][h] >> 16); S.fS[bn][pos] = (short)((u32)S.hbest[2][h] & 0xFFFF);
        }
        nNew += passTot;
    }
    if(nNew > C::WCAP) { if(gl == 0) { st.itersRun = d; st.err = __LINE__; st.needTier = tier_for_frontier(nNew); } DSYNC(); return PH_DONE; }
    DSYNC();
    // reset the hash entries used by this iteration (the survivors live in the frontier buffer now).  A wide frontier's sort borrows the hash's
    // value array, so it resets first; the narrow case keeps the reset at the end of the iteration, off the path to the sorted frontier
    const bool wideSort = C::WCAP > GW && nNew > GW;
    if(wideSort) for(int t = gl; t < nT; t += GW) { int h = S.tlist[t]; S.hkey[h] = HKEY_EMPTY; S.hbest[0][h] = 0; S.hbest[1][h] = 0; S.hbest[2][h] = 0; }
    // ... then put into std::map order (x, y, z) = key order
    if(nNew > 1 && nNew <= GW) {
        // every survivor counts the smaller keys among them (= its rank) and the buffer is rewritten in rank order
        const bool act = gl < nNew;
        u64 key = 0; typename C::Slot vs = 0; short vD = 0, vG = 0, vS = 0; int rank = 0;
        if(act) {
            key = S.fkey[bn][gl]; vs = S.fslot[bn][gl]; vD = S.fD[bn][gl]; vG = S.fG[bn][gl]; vS = S.fS[bn][gl];
            for(int u = 0; u < nNew; u++) rank += (S.fkey[bn][u] < key) ? 1 : 0;
        }
        DSYNC();
        if(act) { S.fkey[bn][rank] = key; S.fslot[bn][rank] = vs; S.fD[bn][rank] = vD; S.fG[bn][rank] = vG; S.fS[bn][rank] = vS; }
    } else if(wideSort) {
        // frontier wider than the group (allele-rich levels: hundreds of cells): bitonic sort of (key, packed payload) pairs in LDS.  The LDS
        // classes sort the frontier buffer in place and borrow the hash's third value array, idle (all zero) between iterations, for the payload;
        // the in-memory class sorts in the block's scratch.  (The rank of every survivor used to be counted against all targets: quadratic,
        // and 99 % of the time of the widest DPs of the Graph M workload.)  Payload: slot (20 bits) and the three scores in 12 bits each
        // (0 = none, else score + 64: a kept cell's scores are >= -22).
        DSYNC();
        int Pn = GW * 2; while(Pn < nNew) Pn <<= 1;              // power of two >= nNew, <= WCAP
        u64* keys; u64* pay;
        if constexpr (C::IN_MEMORY) {
            if(Pn <= DP_SORT_SCRATCH) { keys = sortScratch; pay = sortScratch + DP_SORT_SCRATCH; }
            else { keys = sl.sort_spill(); pay = keys + C::WCAP; }            // wider than the LDS scratch: in the slab (the group fences of this class order global memory)
        } else { keys = &S.fkey[bn][0]; pay = (u64*)&S.hbest[2][0]; }
        auto enc = [](int v) -> u64 { return v == DP_NEG ? 0ull : (u64)(u32)(v + 64); };
        auto dec = [](u64 e) -> short { return e == 0 ? (short)DP_NEG : (short)((int)e - 64); };
        for(int i = gl; i < Pn; i += GW) {
            if(i < nNew) {
                pay[i] = ((u64)(u32)(int)S.fslot[bn][i] << 36) | (enc(S.fD[bn][i]) << 24) | (enc(S.fG[bn][i]) << 12) | enc(S.fS[bn][i]);
                if constexpr (C::IN_MEMORY) keys[i] = S.fkey[bn][i];
            } else { keys[i] = ~0ull; pay[i] = 0; }
        }
        DSYNC();
        for(int k = 2; k <= Pn; k <<= 1)
            for(int j = k >> 1; j > 0; j >>= 1) {
                for(int idx = gl; idx < (Pn >> 1); idx += GW) {
                    const int i = ((idx & ~(j - 1)) << 1) | (idx & (j - 1)), l = i | j;
                    const bool up = (i & k) == 0;
                    const u64 ka = keys[i], kb = keys[l];
                    if((ka > kb) == up) { const u64 pa = pay[i], pb = pay[l]; keys[i] = kb; keys[l] = ka; pay[i] = pb; pay[l] = pa; }
                }
                // (in-memory class: the pairs being sorted sit in LDS unless the frontier is wider than the scratch -- a workgroup barrier orders them; the agent-scope
                //  fences of DSYNC, an L2 write-back and an L1 invalidate per pass, 66 passes for 2 048 pairs, are for state in HBM)
                if constexpr (C::IN_MEMORY) { if(Pn <= DP_SORT_SCRATCH) { if constexpr (GW > 64) blk_barrier(); else { WSYNC(); } } else DSYNC(); }
                else DSYNC();
            }
        for(int i = gl; i < Pn; i += GW) {
            const u64 pv = pay[i];
            if(i < nNew) {
                S.fD[bn][i] = dec((pv >> 24) & 0xFFF); S.fG[bn][i] = dec((pv >> 12) & 0xFFF); S.fS[bn][i] = dec(pv & 0xFFF); S.fslot[bn][i] = (typename C::Slot)(int)(pv >> 36);
                if constexpr (C::IN_MEMORY) S.fkey[bn][i] = keys[i];
            }
            if constexpr (!C::IN_MEMORY) pay[i] = 0;
        }
    }
    DSYNC();
    if(!wideSort) for(int t = gl; t < nT; t += GW) { int h = S.tlist[t]; S.hkey[h] = HKEY_EMPTY; S.hbest[0][h] = 0; S.hbest[1][h] = 0; S.hbest[2][h] = 0; }
    if(gl == 0) {
        st.b2 = b1; st.b1 = bn;                                                               // m2 := m1; m1 := this, :1104-1105
        st.n2 = n1; st.n1 = nNew;
        st.d = d + 1; st.itersRun = d;
        st.nCells = nCells; st.nCompleted = nCompletedNew; st.earlyInit = earlyInit; st.earlyMaxNat = earlyMaxNat;
        st.curMax = curMax; st.lastInc = lastInc; if(firstMaxSlot >= 0) st.firstMaxSlot = firstMaxSlot;
        st.cellsEvaluated += (u32)nT;
#ifdef HLALA_DP_PROFILE
        if(slow) S.pfSlow++; S.pfImp += S.nImp; if(prepass) S.pfPre++; if(nT > S.pfMaxNT) S.pfMaxNT = nT; if(nNew > S.pfMaxF) S.pfMaxF = nNew;
#endif
    }
    DSYNC();
    DP_TQ(2);
    return PH_RUN;
}

// many ties (allele-rich levels: thousands of sequence-complete cells): every tie gets a number whose order is the string order (xz_key), the
// ties are compacted into the slab, and the selectedIndex-th smallest number is found bit by bit.  Register-hungry: only the classes that can hold thousands of complete
// cells (frontier > 64) carry it; in the 16- / 32- / 64-lane kernels it cost 8 more spilled VGPRs in the persistent loop (+16 % kernel time).
template <class C>
__device__ inline int dp_select_many(const DpSlabT<C>& sl, const DevGraph& G, int nCompleted, int best, int selectedIndex)
{
    constexpr int GW = C::GW;
    const int gl = grp_lane<GW>();
    int found = -1;
    int nt = 0;
    for(int i0 = 0; i0 < nCompleted; i0 += GW) {
        int i = i0 + gl; bool tie = false; int s = 0; u64 kk = 0;
        if(i < nCompleted) { s = sl.completed()[i]; if(sl.cell()[s].sc[0] == best) { tie = true; u64 k = sl.cell()[s].key; kk = xz_key(key_x(k), key_node(k) - G.level_off[key_x(k)]); } }
        int tieTot; const int rk = grp_rank<GW>(tie, tieTot);
        if(tie) { const int pos = nt + rk; sl.tie_slot()[pos] = s; sl.tie_key()[pos] = kk; }
        nt += tieTot;
    }
    dp_sync<C>();
    u64 prefix = 0; int k = selectedIndex;
    for(int bit = 51; bit >= 0; bit--) {
        int cnt = 0;
        for(int i = gl; i < nt; i += GW) { const u64 kk = sl.tie_key()[i]; if((kk >> (bit + 1)) == (prefix >> (bit + 1)) && !((kk >> bit) & 1ull)) cnt++; }
        cnt = grp_sum_i32<GW>(cnt);
        if(k >= cnt) { k -= cnt; prefix |= 1ull << bit; }
    }
    for(int i = gl; i < nt; i += GW) if(sl.tie_key()[i] == prefix) found = sl.tie_slot()[i];
    return found;
}

// ---- end cell, :1381-1517
template <class C>
__device__ inline int dp_select(DpLdsT<C>& S, const DpSlabT<C>& sl, const DevGraph& G, u32 rng_seed, const bool fwd)
{
    constexpr int GW = C::GW;
    const int gl = grp_lane<GW>();
    DpState& st = S.st;
    const int nCompleted = guni<GW>(st.nCompleted);
    const u32 seed = rng_seed + (u32)guni<GW>(st.item);
    const int curMax = guni<GW>(st.curMax), firstMaxSlot = guni<GW>(st.firstMaxSlot), start_seq = guni<GW>(st.start_seq);
    int endSlot = -1, endScore = 0;
    if(GW <= 64 && nCompleted > 0 && nCompleted <= GW) {
        // the common case -- at most one sequence-complete cell per lane: slot, score and key are read once (two dependent round trips; the passes below
        // read them three times over, the tie ranking once more per pair of cells) and the ties rank each other through lane shuffles
        int s = -1, sc0 = DP_NEG; u64 k = 0;
        if(gl < nCompleted) { s = sl.completed()[gl]; const CellRec* cr = sl.cell() + s; sc0 = cr->sc[0]; k = cr->key; }
        const int best = grp_max_i32<GW>(sc0);
        const bool tie = gl < nCompleted && sc0 == best;
        const int nTies = grp_sum_i32<GW>(tie ? 1 : 0);
        u32 sd = seed;
        const int selectedIndex = glibc_rand_r(&sd) % nTies;                                // Utilities.cpp:922-927
        int found = -1;
        if(nTies == 1) { if(tie) found = s; }
        else {
            // the tie with exactly `selectedIndex` ties before it in "x/z" string order
            const int kx = key_x(k), kz = tie ? key_node(k) - G.level_off[kx] : 0;
            const int l0 = (int)(lane_id() & ~(u32)(GW - 1));
            int rank = 0;
            for(int u = 0; u < nCompleted; u++) {
                const int ux = __shfl(kx, l0 + u), uz = __shfl(kz, l0 + u), ut = __shfl(tie ? 1 : 0, l0 + u);
                if(ut && u != gl && xz_less(ux, uz, kx, kz)) rank++;
            }
            if(tie && rank == selectedIndex) found = s;
        }
        endSlot = grp_max_i32<GW>(found); endScore = best;
    } else if(nCompleted > 0) {
        int best = DP_NEG;
        for(int i = gl; i < nCompleted; i += GW) best = max(best, (int)sl.cell()[sl.completed()[i]].sc[0]);
        best = grp_max_i32<GW>(best);
        int nTies = 0;
        for(int i = gl; i < nCompleted; i += GW) if(sl.cell()[sl.completed()[i]].sc[0] == best) nTies++;
        nTies = grp_sum_i32<GW>(nTies);
        u32 sd = seed;
        int selectedIndex = glibc_rand_r(&sd) % nTies;                                      // Utilities.cpp:922-927
        // the tie with exactly `selectedIndex` ties before it in "x/z" string order
        int found = -1;
        if(C::WCAP <= 64 || nTies <= 2 * GW) {
            for(int i0 = 0; i0 < nCompleted; i0 += GW) {
                int i = i0 + gl;
                if(i < nCompleted) {
                    int s = sl.completed()[i];
                    if(sl.cell()[s].sc[0] == best) {
                        u64 k = sl.cell()[s].key; int rank = 0;
                        if(nTies > 1) {
                            int kz = key_node(k) - G.level_off[key_x(k)];
                            for(int u = 0; u < nCompleted; u++) { int su = sl.completed()[u]; if(su != s && sl.cell()[su].sc[0] == best) { u64 ku = sl.cell()[su].key;
                                if(xz_less(key_x(ku), key_node(ku) - G.level_off[key_x(ku)], key_x(k), kz)) rank++; } }
                        }
                        if(rank == selectedIndex) found = s;
                    }
                }
            }
        } else if constexpr (C::WCAP > 64) found = dp_select_many<C>(sl, G, nCompleted, best, selectedIndex);
        endSlot = grp_max_i32<GW>(found); endScore = best;
    } else if(curMax > 0) {
        endSlot = firstMaxSlot; endScore = sl.cell()[firstMaxSlot].sc[0];
    }
    if(endSlot < 0) return PH_DONE;                                                          // no extension (have = 0)
    u64 ek = sl.cell()[endSlot].key;
    const int yEnd = key_y(ek);
    if(gl == 0) {
        st.endSlot = endSlot; st.endScore = endScore;
        if(fwd) { st.sb = start_seq; st.se = yEnd - 1; }                                     // toVerboseSeedChain, VirtualNWUnique.cpp:28-29
        else { st.sb = yEnd; st.se = start_seq - 1; }
        S.btSlot = endSlot; S.btM = 0; S.btX = key_x(ek); S.btY = yEnd; S.btGuard = 0; S.btDone = 0; S.nNew = 0; S.nKeepF = 0;
    }
    DSYNC();
    return PH_BT;
}

// ---- backtrace, :1109-1354: lane 0 of the group chases at most `maxSteps` back pointers per call
template <class C>
__device__ inline int dp_backtrace(DpLdsT<C>& S, const DpSlabT<C>& sl, int maxSteps, const bool fwd)
{
    constexpr int GW = C::GW;
    const int gl = grp_lane<GW>();
    DpState& st = S.st;
    const int dir = fwd ? 1 : -1;
    if(gl == 0) {
        const int startLevel = st.startLevel, start_seq = st.start_seq;
        int slot = S.btSlot, m = S.btM, x = S.btX, y = S.btY, guardSteps = S.btGuard;
        int nSteps = S.nNew, nCols = S.nKeepF;
        int done = 0;
        for(int it = 0; it < maxSteps; it++) {
            if(!((x != startLevel || y != start_seq) && nSteps < C::STEPS && guardSteps++ < 4 * C::STEPS)) { done = 1; break; }
            const CellRec* cr = sl.cell() + slot;
            u32 b = cr->bt[m]; const u64 ckey = cr->key;      // (one 32-byte record: both loads are in flight together)
            int kind = bt_kind(b);
            int prev = bt_prev(b);
            int px = 0;
            if(kind == K_JUMP) px = key_x(sl.cell()[prev].key);
            if(kind != K_HOP) {
                int len = 1;
                if(kind == K_JUMP) len = px > x ? px - x : x - px;
                sl.step_bt()[nSteps] = b; sl.step_xy()[nSteps] = ckey; nSteps++; nCols += len;       // the cell the step arrives at: x, y and its node
            }
            if(kind == K_DIAG) { x -= dir; y -= dir; }
            else if(kind == K_GGAP) { y -= dir; }
            else if(kind == K_SGAP) { x -= dir; }
            else if(kind == K_JUMP) { x = px; }
            slot = prev; m = bt_src(b);
        }
        if(!done && !(x != startLevel || y != start_seq)) done = 1;
        S.btSlot = slot; S.btM = m; S.btX = x; S.btY = y; S.btGuard = guardSteps; S.nNew = nSteps; S.nKeepF = nCols;
        if(done) {
            if(nSteps >= C::STEPS || guardSteps >= 4 * C::STEPS) st.err = __LINE__;
            st.nSteps = nSteps; st.nCols = nCols;
        }
        S.btDone = done;
    }
    DSYNC();
    if(guni<GW>(S.btDone)) return guni<GW>(st.err) ? PH_DONE : PH_EXPAND;
    return PH_BT;
}

// ---- all lanes of the group expand the steps into alignment columns, written into the chain's output row:
// the left extension at its final place [seq_begin, seq_begin + n), the right extension right-aligned in the row
// (k_stitch_chains moves it next to the seed once the left extension's length is known)
template <class C>
__device__ inline int dp_expand(DpLdsT<C>& S, const DpSlabT<C>& sl, const DevGraph& G, const DevBatch& B, const int4* nrec, const bool fwd)
{
    constexpr int GW = C::GW;
    const int gl = grp_lane<GW>();
    DpState& st = S.st;
    const int nSteps = guni<GW>(st.nSteps), nCols = guni<GW>(st.nCols);
    const int stride = B.stride;
    const int max_seqI = guni<GW>(st.seqLen);
    const int sb = guni<GW>(st.sb), se = guni<GW>(st.se);
    const uint8_t* seqp = B.read_bases + guni<GW>(st.rOff);
    if(nCols > stride) { if(gl == 0) st.err = -1000000 - nCols; DSYNC(); return PH_DONE; }
    if(sb > se) { if(gl == 0) { st.err = __LINE__; st.have = 0; } DSYNC(); return PH_DONE; }
    if(gl == 0) st.have = 1;
    DSYNC();
    const int c = guni<GW>(st.item) >> 1;
    const int rowOff = fwd ? stride - nCols : sb;
    if(rowOff + nCols > stride) return PH_DONE;             // the stitched chain cannot fit the row: k_stitch_chains reports the column error
    const size_t cb = row_base(B, c) + rowOff;
    int* oL = B.ext_level + cb; int* oE = B.ext_edge + cb; uint8_t* oG = B.ext_g + cb; uint8_t* oS = B.ext_s + cb;
    int base = 0;
    for(int s0 = 0; s0 < nSteps; s0 += GW) {
        int s = s0 + gl; bool act = s < nSteps;
        u32 b = act ? sl.step_bt()[s] : 0; u64 xy = act ? sl.step_xy()[s] : 0;
        int kind = bt_kind(b);
        // resolve the graph object behind the rank j: the (j+1)-th edge of the previous cell's node that leads to this cell's node, or the (j+1)-th
        // entry of its jump table that does (degrees beyond two are rare and only the cells of the chosen path come here)
        // (the previous node's 32-byte record -- first edge, degree, the targets and labels of its first two edges, its first jump -- answers the common case
        // in one round trip; the CSR offsets, the scan of the targets and the label of the edge were three dependent ones)
        int pnode = 0, robj = -1, recLab = -1;
        if(act && kind != K_GGAP) {
            u64 pkey = sl.cell()[bt_prev(b)].key; pnode = key_node(pkey);
            const int node = key_node(xy);
            int j = bt_edge(b);
            const int4 r0 = nrec[2 * (size_t)pnode], r1 = nrec[2 * (size_t)pnode + 1];
            if(kind == K_JUMP) {
                const int* jn = fwd ? G.jf_node : G.jb_node;
                const int nj = (int)((u32)r0.y >> 16); int q = r1.x; const int q1 = q + nj;
                if(!(nj > 0 && r1.y == node && j == 0)) for(; q < q1; q++) if(jn[q] == node && j-- == 0) break;
                if(q < q1) robj = (fwd ? G.jf_path : G.jb_path)[q];
            } else {
                const int* et = fwd ? G.out_to : G.in_from;
                const int deg = r0.y & 0xFFFF; int q = r0.x; const int q1 = q + deg;
                if(deg > 0 && r0.z == node && j == 0) recLab = r1.w & 0xFF;
                else if(deg > 1 && r0.w == node && j == ((r1.w >> 16) & 1)) { q++; recLab = (r1.w >> 8) & 0xFF; }
                else for(; q < q1; q++) if(et[q] == node && j-- == 0) break;
                if(q < q1) robj = (fwd ? G.out_eid : G.in_eid)[q];
            }
            if(robj < 0) { st.err = __LINE__; act = false; }        // (cannot happen: the rank was derived from these very arrays)
        }
        int len = act ? (kind == K_JUMP ? G.path_len[robj] : 1) : 0;
        int total; int off = grp_excl_scan<GW>(len, total);
        if(act) {
            int x = key_x(xy), y = key_y(xy);
            int start = fwd ? (nCols - (base + off) - len) : (base + off);     // forward traces are reversed at the end, :1319-1326
            if(kind == K_JUMP) {                                                       // :1282-1307
                int p = robj; long long po = G.path_off[p];
                int lvl0 = G.node_level[G.edge_from_new[G.path_edges[po]]];
                for(int j = 0; j < len; j++) { oL[start + j] = lvl0 + j; oE[start + j] = G.path_edges[po + j]; oG[start + j] = '_'; oS[start + j] = '_'; }
            } else {
                int eid = robj;
                unsigned char sc = fwd ? (y >= 1 ? seqp[y - 1] : 0) : (y < max_seqI ? seqp[y] : 0);
                int lvl = fwd ? x - 1 : x;
                const unsigned char gch = kind == K_GGAP ? (unsigned char)'_' : (recLab >= 0 ? (unsigned char)recLab : G.edge_label[eid]);
                if(kind == K_DIAG) { oL[start] = lvl; oE[start] = eid; oG[start] = gch; oS[start] = sc; }
                else if(kind == K_GGAP) { oL[start] = -1; oE[start] = -1; oG[start] = '_'; oS[start] = sc; }
                else { oL[start] = lvl; oE[start] = eid; oG[start] = gch; oS[start] = '_'; }
            }
        }
        base += total;
    }
    return PH_DONE;
}

// ------------------------------------------------------------------------------------------
// Input checks of extendSeedChain (extensionAligner.cpp:184-319) per chain and the list of DP items.
// work_counter[8] / [9] count the left / right items; chains without any usable seed get their final status here.
// The DP item of chain c in direction d (0 = left, 1 = right), extensionAligner.cpp:220-319; false: no such DP.
__device__ inline bool dp_item_for(const DevGraph& G, const DevBatch& B, int c, int d, DpItem& it)
{
    if(B.unpaired) return false;            // alignOneLongRead only pads the seed chain (extendToFullSequenceLength, processBAM.cpp:3733-3735)
    if(B.seed_status[c] != HLALA_CHAIN_OK) return false;
    const int r = B.chain_read[c];
    const int rOff = B.read_off[r], seqLen = B.read_off[r + 1] - rOff;
    const int nSeed = B.seed_ncols[c], sBegin = B.seed_begin[c], sEnd = B.seed_end[c];
    if(seqLen > DP_SEQCAP || seqLen < 1 || nSeed < 1 || sBegin < 0 || sEnd >= seqLen || sBegin > sEnd) return false;
    const size_t cb = row_base(B, c);
    const int e0 = B.seed_edge[cb], e1 = B.seed_edge[cb + nSeed - 1];
    if(e0 < 0 || e1 < 0 || e0 >= G.E || e1 >= G.E) return false;
    it.rOff = rOff; it.seqLen = seqLen; it.pad0 = 0; it.pad1 = 0; it.item = 2 * c + d;
    if(d == 0) {                                                                        // left extension, extensionAligner.cpp:220-268
        if(sBegin == 0) return false;
        const int firstNode = G.edge_from_new[e0]; const int lvl = G.node_level[firstNode];
        if(lvl <= 0) return false;
        it.start_seq = sBegin; it.startLevel = lvl; it.startNode = firstNode;
    } else {                                                                            // right extension, :271-319
        if(sEnd == seqLen - 1) return false;
        const int lastNode = G.edge_to_new[e1]; const int lvl = G.node_level[lastNode];
        if(lvl >= G.L - 1) return false;
        it.start_seq = sEnd + 1; it.startLevel = lvl; it.startNode = lastNode;
    }
    return true;
}

__global__ void k_dp_items(const DevGraph* __restrict__ Gp, const DevBatch* __restrict__ Bp, DpItem* items)
{
    const DevGraph& G = *Gp;
    const DevBatch& B = *Bp;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    // ---- every chain: DP outputs reset; chains that did not pass stage A get their final status here (dp_alias_head / dp_alias_next: memset by the host)
    if(t < B.n_chains) {
        const int st = B.seed_status[t];
        B.dp_iters[2 * t] = 0; B.dp_iters[2 * t + 1] = 0; B.dp_score[2 * t] = INT32_MIN; B.dp_score[2 * t + 1] = INT32_MIN;
        B.dp_ncols[2 * t] = -1; B.dp_ncols[2 * t + 1] = -1; B.dp_err[2 * t] = 0; B.dp_err[2 * t + 1] = 0;
        if(st != HLALA_CHAIN_OK) {
            B.ext_status[t] = st; B.ext_ncols[t] = 0;
            if(st < 0) atomicAdd(&B.counters[CNT_ERRORS], 1ull);
        }
    }
    // ---- the item lists: slot t of the left list and slot t of the right list belong to the t-th chain IN POSITION ORDER (kernel_order.hip; input order
    // without one), a slot without a DP holds item = -1.  The DP classes draw slots in list order, so the DP calls in flight at one time start on
    // neighbouring levels and read neighbouring node records.  (A dense list appended to by the waves in launch order is sorted only as far as the grid
    // runs in order: a twelfth of the graph at a time.)
    const int nOrd = ordered_chains(B);
    const int c = t < nOrd ? (B.chain_order ? B.chain_order[t] : t) : -1;
    bool needL = false, needR = false;
    int nShared = 0;
    DpItem itL, itR;
    if(c >= 0 && B.seed_status[c] == HLALA_CHAIN_OK) {
        const int stride = B.stride;
        const int r = B.chain_read[c];
        const int rOff = B.read_off[r], seqLen = B.read_off[r + 1] - rOff;
        const size_t cb = row_base(B, c);
        const int nSeed = B.seed_ncols[c], sBegin = B.seed_begin[c], sEnd = B.seed_end[c];
        int err = 0;
        if((!B.unpaired && seqLen > DP_SEQCAP) || seqLen < 1 || nSeed < 1 || sBegin < 0 || sEnd >= seqLen || sBegin > sEnd) err = HLALA_CHAIN_ERR_INPUT;     // DP_SEQCAP bounds the DP only
        if(!err) {
            const int e0 = B.seed_edge[cb], e1 = B.seed_edge[cb + nSeed - 1];
            if(e0 < 0 || e1 < 0 || e0 >= G.E || e1 >= G.E) err = HLALA_CHAIN_ERR_INPUT;
        }
        if(err) { B.ext_status[c] = err; B.ext_ncols[c] = 0; B.ext_ll[c] = 0.0; atomicAdd(&B.counters[CNT_ERRORS], 1ull); }
        else {
            B.ext_status[c] = EXT_PENDING;
            needL = dp_item_for(G, B, c, 0, itL);
            needR = dp_item_for(G, B, c, 1, itR);
            // The iterations of a DP are a function of (read, direction, start cell); the chain's random seed only enters when the end
            // cell is drawn among equal ones.  Alignments of one read to homologous contigs project onto the same graph cells all the
            // time: such a DP runs once, for the lowest chain of the read that needs it, and the group that ran it then repeats the
            // end-cell choice (with the other chain's seed), the backtrace and the expansion for every chain linked to it here.
            if(!B.from_seeds && (needL || needR)) {
                for(int c2 = B.chain_off[r]; c2 < c && (needL || needR); c2++) {
                    DpItem o;
                    if(needL && dp_item_for(G, B, c2, 0, o) && o.start_seq == itL.start_seq && o.startNode == itL.startNode) { B.dp_alias_next[2 * c] = atomicExch(&B.dp_alias_head[2 * c2], 2 * c); needL = false; nShared++; }
                    if(needR && dp_item_for(G, B, c2, 1, o) && o.start_seq == itR.start_seq && o.startNode == itR.startNode) { B.dp_alias_next[2 * c + 1] = atomicExch(&B.dp_alias_head[2 * c2 + 1], 2 * c + 1); needR = false; nShared++; }
                }
            }
        }
    }
    // Jump-free calls (DpTinyJF): no node of the levels a call can reach has a gap-path jump (FlatGraph::jfree_out / jfree_in).  A heuristic bound: a call
    // that meets a jump anyway is re-run in the next class.
    bool jfL = false, jfR = false;
    // (reach: a frontier cell lies at most as many levels from the start as the read has bases left, plus the five or six levels of gaps the X-drop window of 15 admits
    //  at 6 + 2 per level; DP_JF_MARGIN on top.  B.dp_jf = the margin + 1, 0 = off.)
    if(needL && B.dp_jf) { const int reach = itL.start_seq + B.dp_jf - 1; jfL = reach < 255 && (int)G.jfree_in[itL.startLevel] > reach; }
    if(needR && B.dp_jf) { const int reach = itR.seqLen - itR.start_seq + B.dp_jf - 1; jfR = reach < 255 && (int)G.jfree_out[itR.startLevel] > reach; }
    // Band calls (kernel_dp_band.hip): the levels the call can reach -- read bases left + dp_band - 1 -- are a run of LINEAR steps (FlatGraph::lin_out / lin_in: one
    // node per level, one edge per step, no '_' label, no jump).  A call that walks further anyway fails over to the general list.
    bool bdL = false, bdR = false;
    int runL = 0, runR = 0;
    int clsL = jfL ? 1 : 0, clsR = jfR ? 1 : 0;       // the list an item goes to: 0 general, 1 jump-free, 2 / 3 / 4 band kernel with 16 / 32 / 64 lanes per call
    // (levels a call is taken to reach: while it has read bases left a cell sits at most dp_band - 1 levels beyond them; afterwards the tail -- one sequence gap per
    //  iteration from the best complete cell, score <= 2 * bases, -6 then -2 per level down to the threshold of -16, at most 40 iterations -- walks up to bases + 6 more)
    if(needL && B.dp_band) { const int jm = itL.start_seq, reach = jm + B.dp_band - 1; runL = (int)G.lin_in[itL.startLevel]; bdL = jm <= BAND_MAXJ64 && runL >= (B.dp_band_risky ? jm : reach + min(jm + 6, 40)); if(bdL) clsL = jm <= BAND_MAXJ16 ? 2 : (jm <= BAND_MAXJ32 ? 3 : 4); }
    if(needR && B.dp_band) { const int jm = itR.seqLen - itR.start_seq, reach = jm + B.dp_band - 1; runR = (int)G.lin_out[itR.startLevel]; bdR = jm <= BAND_MAXJ64 && runR >= (B.dp_band_risky ? jm : reach + min(jm + 6, 40)); if(bdR) clsR = jm <= BAND_MAXJ16 ? 2 : (jm <= BAND_MAXJ32 ? 3 : 4); }
    // Two-track band calls (kernel_dp_band2.hip, round 6): not linear, but every level the call is taken to reach -- read bases left + dp_band2 - 1 -- holds one or two nodes
    // of at most four edges (FlatGraph::trk_out / trk_in); the kernel itself finds the one gap-path jump it can take among them.  A call that walks further fails over.
    if(needL && !bdL && B.dp_band2) { const int jm = itL.start_seq, run = (int)G.trk_in[itL.startLevel]; if(jm <= B.dp_band2_maxj && run >= min(jm + B.dp_band2 - 1, 255)) { clsL = jm <= B2_MAXJ16 ? 5 : (jm <= B2_MAXJ32 ? 6 : 7); runL = run; } }
    if(needR && !bdR && B.dp_band2) { const int jm = itR.seqLen - itR.start_seq, run = (int)G.trk_out[itR.startLevel]; if(jm <= B.dp_band2_maxj && run >= min(jm + B.dp_band2 - 1, 255)) { clsR = jm <= B2_MAXJ16 ? 5 : (jm <= B2_MAXJ32 ? 6 : 7); runR = run; } }
    if(t < nOrd) {
        int4* sl = (int4*)(items + t); int4* sr = (int4*)(items + (size_t)B.n_chains + t);
        if(needL) { sl[0] = make_int4(itL.item, itL.rOff, itL.seqLen, itL.start_seq); sl[1] = make_int4(itL.startLevel, itL.startNode, clsL, runL); } else sl[0] = make_int4(-1, 0, 0, 0);
        if(needR) { sr[0] = make_int4(itR.item, itR.rOff, itR.seqLen, itR.start_seq); sr[1] = make_int4(itR.startLevel, itR.startNode, clsR, runR); } else sr[0] = make_int4(-1, 0, 0, 0);
    }
    // ---- how many items of each of the ten lists (three band lists, jump-free, general; left / right each) this block holds: k_order_scan turns the counts of all
    // blocks into the blocks' places in the dense lists, k_dp_lists writes the slot numbers there -- in position order, nothing is left to the atomics.
    // work_counter[8] / [9]: left / right DP calls of the batch, [6]: the jump-free ones among them, [WC_BAND_CALLS]: the band ones (statistics)
    __shared__ int blkCnt[DPL_N];
    if(threadIdx.x < DPL_N) blkCnt[threadIdx.x] = 0;
    __syncthreads();
    const int lane = lane_id();
    nShared = wave_sum_i32(nShared);
    if(lane == 0 && nShared) atomicAdd(&B.counters[CNT_DP_SHARED], (u64)nShared);
    const u64 mL = __ballot(needL), mR = __ballot(needR);
    const int kL = needL ? dpl_of_class(clsL) : -1, kR = needR ? dpl_of_class(clsR) + 1 : -1;
    const u64 mJL = __ballot(needL && clsL == 1), mJR = __ballot(needR && clsR == 1), mBL = __ballot(needL && clsL >= 2 && clsL <= 4), mBR = __ballot(needR && clsR >= 2 && clsR <= 4);
    const u64 m2L = __ballot(needL && clsL >= 5), m2R = __ballot(needR && clsR >= 5);
    if(lane == 0) {
        if(mL) atomicAdd(&B.work_counter[8], __popcll(mL));
        if(mR) atomicAdd(&B.work_counter[9], __popcll(mR));
        if(mJL | mJR) atomicAdd(&B.work_counter[6], __popcll(mJL) + __popcll(mJR));
        if(mBL | mBR) atomicAdd(&B.work_counter[WC_BAND_CALLS], __popcll(mBL) + __popcll(mBR));
        if(m2L | m2R) atomicAdd(&B.work_counter[WC_B2_CALLS], __popcll(m2L) + __popcll(m2R));
    }
#pragma unroll
    for(int k = 0; k < DPL_N; k++) {
        const u64 mk = __ballot((k & 1) ? kR == k : kL == k);
        if(lane == 0 && mk) atomicAdd(&blkCnt[k], __popcll(mk));
    }
    __syncthreads();
    if(threadIdx.x < DPL_N) B.dp_blk[(size_t)threadIdx.x * gridDim.x + blockIdx.x] = blkCnt[threadIdx.x];
}

// The ten dense item lists of the first DP classes: list k (DPL_BAND16 / DPL_BAND32 / DPL_BAND64 / DPL_JF / DPL_GEN, + 1 for the right extensions) occupies
// dp_list[dp_blk[k * nBlk] .. dp_blk[(k + 1) * nBlk]) -- dp_blk after its exclusive scan: entry k * nBlk + b = where block b's items of list k start --, its
// entries are the slots of k_dp_items' item arrays in position order.  Same grid as k_dp_items.
__global__ void k_dp_lists(const DevBatch* __restrict__ Bp, const DpItem* __restrict__ items)
{
    const DevBatch& B = *Bp;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int nOrd = ordered_chains(B);
    int kL = -1, kR = -1;             // list of the left / right item of this slot
    if(t < nOrd) {
        const int4* sl = (const int4*)(items + t); const int4* sr = (const int4*)(items + (size_t)B.n_chains + t);
        if(sl[0].x >= 0) kL = dpl_of_class(sl[1].z);
        if(sr[0].x >= 0) kR = dpl_of_class(sr[1].z) + 1;
    }
    __shared__ int waveCnt[DPL_N][4];      // [list][wave of the block]
    const int lane = lane_id(), wv = (int)(threadIdx.x >> 6);
    u64 m[DPL_N];
#pragma unroll
    for(int k = 0; k < DPL_N; k++) m[k] = __ballot((k & 1) ? kR == k : kL == k);
    if(lane == 0) {
#pragma unroll
        for(int k = 0; k < DPL_N; k++) waveCnt[k][wv] = (int)__popcll(m[k]);
    }
    __syncthreads();
    const u64 below = (1ull << lane) - 1ull;
#pragma unroll
    for(int k = 0; k < DPL_N; k++) {
        const bool mine = (k & 1) ? kR == k : kL == k;
        if(mine) {
            int before = 0;
            for(int w2 = 0; w2 < wv; w2++) before += waveCnt[k][w2];
            const int pos = B.dp_blk[(size_t)k * gridDim.x + blockIdx.x] + before + (int)__popcll(m[k] & below);
            B.dp_list[pos] = (k & 1) ? B.n_chains + t : t;
        }
    }
}

// ------------------------------------------------------------------------------------------
// TIER 0: items of k_dp_items, the left extensions first and then the right extensions, so that the direction (and with it
//         every choice between the out- and the in-edge arrays) is uniform across the four groups of a wavefront;
// TIER k > 0: items that outgrew the class of tier k-1 (retry list k).
// Round 6: the classes from DP_POOL_TIER on (broad, large, in-memory: a few hundred long calls per batch whose cost is the latency of their slowest call, paid per
// LAUNCH) take the lists of SEVERAL batches in one launch -- hlala_align_batch with a tail pool (hlala_set_tail_pool) leaves them pending and runs them once for up to
// DP_POOL_MAX consecutive batches of a sample.  The kernel walks the batches one after the other: every list has its own fetch counter, a block that finds a list
// drained goes on to the next batch's.
struct DpPoolArgs {
    int n;                                      // batches of this launch (1: the batch of the plain arguments)
    u32 seed[DP_POOL_MAX];                      // rng_seed + 2 * first chain of the batch
    const DevBatch* B[DP_POOL_MAX];
    const DpItem* items[DP_POOL_MAX];
    const uint8_t* bases[DP_POOL_MAX];
};

template <class C, int TIER>
__device__ __forceinline__ void dp_class_run(const DevGraph& G, const DevBatch& B, const DpItem* __restrict__ items, DpLdsT<C>& S, DpSlabT<C> sl, u64* sortScratch, const u32 rng_seed,
                                             const int4* __restrict__ nrecOut, const int4* __restrict__ nrecIn, const uint8_t* __restrict__ readBases);

template <class C, int TIER>
__global__ __launch_bounds__(C::THREADS, C::WAVES) void k_dp(const DevGraph* __restrict__ Gp, const DevBatch* __restrict__ Bp, const DpItem* __restrict__ items,
                                                        char* slabs, size_t slabBytes, u32 rng_seed,
                                                        // the arrays of the inner loop are passed as kernel arguments: pointers loaded from the descriptors are generic
                                                        // (flat_load, which also ties up the LDS counter), kernel-argument pointers are known to be global
                                                        const int4* __restrict__ nrecOut, const int4* __restrict__ nrecIn, const uint8_t* __restrict__ readBasesArg, const DpPoolArgs pool)
{
    constexpr int GW = C::GW;
    constexpr int NG = GW >= 64 ? 1 : 64 / GW;           // DPs per block: groups of a wavefront, or one DP for the whole block
    static_assert(C::THREADS == (GW >= 64 ? GW : 64), "block size");
    // graph / batch descriptors stay in memory (scalar loads on demand): passing them by value costs ~150 SGPRs
    const DevGraph& G = *Gp;
    const int g = (int)((threadIdx.x & 63) / GW);
    DpLdsT<C>* Sp; DpSlabT<C> sl; u64* sortScratch = nullptr;
    if constexpr (C::IN_MEMORY) {
        __shared__ u64 scratch[2 * DP_SORT_SCRATCH];
        char* base = slabs + (size_t)blockIdx.x * slabBytes;          // [ structure | scratch slab ]
        Sp = (DpLdsT<C>*)base; sl.base = base + ((sizeof(DpLdsT<C>) + 255) & ~(size_t)255); sortScratch = scratch;
    } else {
        __shared__ DpLdsT<C> SS[NG];
        Sp = &SS[g]; sl.base = slabs + ((size_t)blockIdx.x * NG + g) * slabBytes;
    }
    if constexpr (TIER >= DP_POOL_TIER) {
        const int nB = __builtin_amdgcn_readfirstlane(pool.n);
        for(int bi = 0; bi < nB; bi++) {
            dp_class_run<C, TIER>(G, *pool.B[bi], pool.items[bi], *Sp, sl, sortScratch, pool.seed[bi], nrecOut, nrecIn, pool.bases[bi]);
            DSYNC();
        }
    } else dp_class_run<C, TIER>(G, *Bp, items, *Sp, sl, sortScratch, rng_seed, nrecOut, nrecIn, readBasesArg);
}

// one class over the lists of one batch (the body of k_dp)
template <class C, int TIER>
__device__ __forceinline__ void dp_class_run(const DevGraph& G, const DevBatch& B, const DpItem* __restrict__ items, DpLdsT<C>& S, DpSlabT<C> sl, u64* sortScratch, const u32 rng_seed,
                                             const int4* __restrict__ nrecOut, const int4* __restrict__ nrecIn, const uint8_t* __restrict__ readBases)
{
    constexpr int GW = C::GW;
    constexpr int NG = GW >= 64 ? 1 : 64 / GW;           // DPs per block: groups of a wavefront, or one DP for the whole block
#ifndef HLALA_DP_DRAW0
#define HLALA_DP_DRAW0 8
#endif
    constexpr int DRAW = (TIER == 0 && C::JF) ? HLALA_DP_DRAW0 : 1;      // items a group draws per atomic: several for the short, even calls of the jump-free list (36.7 -> 32.5 ms at 3 M calls); the general list and the later
                                                                          // classes hold long calls that sit together in position order -- draws of 8 there cost 9 ms of tail (profiles/r05_experiments.txt)
    const int gl = grp_lane<GW>();

    if(gl == 0) { S.accCalls = 0; S.accIters = 0; S.accCells = 0; S.accEdges = 0; }
#ifdef HLALA_DP_TIMING
    if(gl == 0) { for(int i = 0; i < 16; i++) S.tPh[i] = 0; }
#endif
#ifdef HLALA_DP_TIMING                                     // build-time switch: cycles per state of the persistent loop -> counters[8..15]
    long long tAcc[6] = {0, 0, 0, 0, 0, 0}; long long trips = 0, runGroups = 0; long long tMark = clock64();
#define DP_T(i) do { long long t_ = clock64(); tAcc[i] += t_ - tMark; tMark = t_; } while(0)
#else
#define DP_T(i) do { } while(0)
#endif

    // TIER 0, general instantiation: after its own two lists the fail-over lists of the band kernel and of the jump-free instantiation (passes 2 and 3)
    constexpr int NPASS = (TIER == 0 && !C::JF) ? 4 : 2;
    for(int pass = 0; pass < NPASS; pass++) {
        const int dirPass = pass & 1;
        const bool foPass = pass >= 2;
        // work_counter: [8]/[9] left / right item counts, [1]/[10] their fetch counters;
        // retry list of tier k = 1..6 and direction p: count [12 + 4(k-1) + 2p], fetched [13 + 4(k-1) + 2p], entries retry_list[(2(k-1) + p) n_chains ...]
        int* fetchCounter = foPass ? &B.work_counter[WC_FO_FETCH + dirPass]
                          : &B.work_counter[TIER == 0 ? (C::JF ? 4 + dirPass : (dirPass ? 10 : 1)) : 13 + 4 * (TIER - 1) + 2 * dirPass];        // [4]/[5]: the jump-free lists
        // (TIER 0 draws from the dense lists of k_dp_lists -- jump-free or general, left or right --, the later tiers from the retry lists)
        const int seg = (C::JF ? DPL_JF : DPL_GEN) + dirPass;
        const bool dense = TIER == 0 && !foPass;
        const int segStart = dense ? uni(B.dp_blk[(size_t)seg * B.dp_nblk]) : 0;
        const int nItems = dense ? uni(B.dp_blk[(size_t)(seg + 1) * B.dp_nblk]) - segStart
                         : (foPass ? uni(B.work_counter[WC_FO_COUNT + dirPass]) : uni(B.work_counter[12 + 4 * (TIER > 0 ? TIER - 1 : 0) + 2 * dirPass]));
        const int* srcList = dense ? B.dp_list + segStart
                           : (foPass ? B.retry_list + (size_t)(14 + dirPass) * (size_t)B.n_chains
                           : B.retry_list + (size_t)(2 * (TIER > 0 ? TIER - 1 : 0) + dirPass) * (size_t)B.n_chains);
        const bool fwd = dirPass != 0;                     // left extensions run backwards (alignerBase: extensionAligner.cpp:229-241)
        if constexpr (DRAW > 1) { if(gl == 0) { S.chunkNext = 0; S.chunkEnd = 0; } DSYNC(); }
        int phase = PH_IDLE;
        bool more = true;
        int edgesAcc = 0;                                  // per-lane partial sum of the edges the current DP touched

        // One trip: every group advances through as many states as it can -- iterate, and on the last iteration straight on to end cell,
        // backtrace (up to DP_BT_STEPS_PER_TRIP pointers), expansion, bookkeeping and the next item -- so that a group spends its trips
        // iterating, not changing state (the states used to be visited in the opposite order: one state change per trip).
        for(;;) {
            if(phase == PH_RUN) phase = dp_iterate<C>(S, sl, G, fwd ? nrecOut : nrecIn, readBases, fwd, edgesAcc, sortScratch);
            DP_T(5);
            if(phase == PH_SELECT) phase = dp_select<C>(S, sl, G, rng_seed, fwd);
            DP_T(4);
            if(phase == PH_BT) phase = dp_backtrace<C>(S, sl, DP_BT_STEPS_PER_TRIP, fwd);
            DP_T(3);
            if(phase == PH_EXPAND) phase = dp_expand<C>(S, sl, G, B, fwd ? nrecOut : nrecIn, fwd);
            DP_T(2);
            if(phase == PH_DONE) {
                // final bookkeeping of this DP in this class; a DP that outgrew the class is queued for the next one and leaves no trace
                const int edges = grp_sum_i32<GW>(edgesAcc);
                if(gl == 0) {
                    DpState& st = S.st;
                    S.nextPhase = PH_IDLE;
#ifdef HLALA_DP_PROFILE
                    if(B.dbg && TIER == HLALA_DP_PROFILE && !C::JF) {
                        const long long cyc = clock64() - S.pfStart;
                        if(cyc > (1ll << HLALA_DP_PROFILE_LOG2) && (HLALA_DP_PROFILE_LOG2 >= 12 || (((u32)st.item * 2654435761u) >> 24) == 0)) { int q = atomicAdd(&B.dbg[0], 1); if(q < 500) { int* r = B.dbg + 16 + 16 * q; r[0] = st.item; r[1] = st.itersRun; r[2] = (int)st.cellsEvaluated; r[3] = st.nCells;
                            r[4] = S.pfSlow; r[5] = S.pfImp; r[6] = S.pfPre; r[7] = (int)(cyc >> 10); r[8] = S.pfMaxNT;
                            r[9] = (int)(S.pfPh[4] >> 10); r[10] = (int)(S.pfPh[5] >> 10); r[11] = (int)(S.pfPh[0] >> 10); r[12] = (int)(S.pfPh[3] >> 10); r[13] = (int)(S.pfPh[6] >> 10); r[14] = (int)(S.pfPh[1] >> 10); r[15] = (int)(S.pfPh[2] >> 10); } }
                    }
#endif
                    const bool capacity = st.err != 0 && st.err > -1000000;
                    bool handed = false;
                    if constexpr (C::JF && TIER == 0) if(capacity && S.jumpMet) {
                        // the jump-free instantiation met a gap-path jump: the call goes to the fail-over list of the GENERAL 16-lane instantiation, not to the 32-lane class
                        if(st.isAlias) ((int*)(items + st.itemIdx))[0] = st.item;
                        const int q = atomicAdd(&B.work_counter[WC_FO_COUNT + dirPass], 1); B.retry_list[(size_t)(14 + dirPass) * (size_t)B.n_chains + q] = st.itemIdx;
                        atomicAdd(&B.work_counter[WC_JF_FAILED], 1);
                        handed = true;
                    }
                    if(handed) { }
                    else if(capacity && TIER < DP_LAST_TIER) {
                        // next tier, or straight to the first tier whose class holds what overflowed (no point in failing again on the way)
                        int to = TIER + 1; if(st.needTier > to) to = st.needTier; if(to > DP_LAST_TIER) to = DP_LAST_TIER;
                        int* cnt = &B.work_counter[12 + 4 * (to - 1) + 2 * dirPass]; int* lst = B.retry_list + (size_t)(2 * (to - 1) + dirPass) * (size_t)B.n_chains;
                        // (a linked duplicate whose own backtrace outgrew the class takes over the item entry: the DP that ran is done with it)
                        if(st.isAlias) ((int*)(items + st.itemIdx))[0] = st.item;
                        int q = atomicAdd(cnt, 1); lst[q] = st.itemIdx;
                        // (all chains that share this DP belong to one read: its pair waits for the side-stream classes)
                        if(to >= DP_SIDE_TIER && B.n_pairs > 0 && !B.unpaired) B.pair_deferred[B.chain_read[st.item >> 1] >> 1] = 1;
                    } else {
                        const int item = st.item;
                        B.dp_iters[item] = st.itersRun; B.dp_score[item] = st.have ? st.endScore : INT32_MIN;
                        B.dp_ncols[item] = st.have ? st.nCols : -1; B.dp_sb[item] = st.sb; B.dp_se[item] = st.se; B.dp_err[item] = st.err;
                        S.accCalls++; S.accIters += (u64)st.itersRun; S.accCells += (u64)st.cellsEvaluated; S.accEdges += (u64)edges;
                        // the chains of the read whose DP starts from the same cell (k_dp_items): same iterations, their own end-cell draw
                        int nx = B.dp_alias_head[item]; if(nx < 0) nx = B.dp_alias_next[item];     // the DP that ran heads the list, a duplicate is on it
                        const bool failed = st.err != 0 && st.err > -1000000;       // capacity of the last class: the same for every copy
                        while(nx >= 0 && failed) {
                            B.dp_iters[nx] = st.itersRun; B.dp_score[nx] = INT32_MIN; B.dp_ncols[nx] = -1; B.dp_sb[nx] = st.sb; B.dp_se[nx] = st.se; B.dp_err[nx] = st.err;
                            S.accCalls++; S.accIters += (u64)st.itersRun; S.accCells += (u64)st.cellsEvaluated; S.accEdges += (u64)edges;
                            nx = B.dp_alias_next[nx];
                        }
                        if(nx >= 0) {
                            st.item = nx; st.isAlias = 1;
                            st.endSlot = -1; st.endScore = 0; st.nSteps = 0; st.nCols = 0; st.have = 0; st.sb = 0; st.se = -1; st.err = 0;
                            S.nextPhase = PH_SELECT;
                        }
                    }
                }
                DSYNC();
                phase = guni<GW>(S.nextPhase);
            }
            DP_T(1);
            if(phase == PH_IDLE && more) {
                // items are drawn DRAW at a time: one word hands out ~88 draws per microsecond (MI355X_MICROARCH.md: dequeue), and the first class asks for
                // 3.8 M items per million pairs -- with one atomic per item the jump-free instantiation ran at exactly that rate (79-83 calls per microsecond
                // whatever the mix of calls: profiles/r05_experiments.txt)
                int w = 0;
                if(gl == 0) {
                    if constexpr (DRAW > 1) {
                        int nx = S.chunkNext;
                        if(nx >= S.chunkEnd) { nx = atomicAdd(fetchCounter, DRAW); S.chunkEnd = nx + DRAW; }
                        S.chunkNext = nx + 1; w = nx;
                    } else w = atomicAdd(fetchCounter, 1);
                }
                w = grp_bcast0<GW>(w);
                if(w >= nItems) more = false;
                else {
                    const int idx = guni<GW>(srcList[w]);
                    const int4* ip = (const int4*)(items + idx);
                    int4 a = ip[0], b = ip[1];
                    DpItem it; it.item = a.x; it.rOff = a.y; it.seqLen = a.z; it.start_seq = a.w; it.startLevel = b.x; it.startNode = b.y; it.pad0 = 0; it.pad1 = 0;
                    phase = dp_begin<C>(S, sl, G, it, idx);
                    edgesAcc = 0;
                }
            }
            DP_T(0);
            if(!__ballot(phase != PH_IDLE)) break;       // every group is idle and found no more work
#ifdef HLALA_DP_TIMING
            trips++; runGroups += __popcll(__ballot(phase == PH_RUN)) / GW;
#endif
            DP_T(5);
        }
    }
    if(gl == 0 && S.accCalls) {
        atomicAdd(&B.counters[CNT_DP_CALLS], S.accCalls); atomicAdd(&B.counters[CNT_DP_ITERS], S.accIters);
        atomicAdd(&B.counters[CNT_DP_CELLS], S.accCells); atomicAdd(&B.counters[CNT_EDGES], S.accEdges);
    }
#ifdef HLALA_DP_TIMING
    if(TIER == 0 && (threadIdx.x & 63) == 0) { for(int i = 0; i < 8; i++) atomicAdd(&B.counters[24 + i], (u64)S.tPh[i]); for(int i = 0; i < 8; i++) atomicAdd(&B.counters[16 + i], (u64)S.tPh[8 + i]);
        for(int i = 0; i < 6; i++) atomicAdd(&B.counters[8 + i], (u64)tAcc[i]); atomicAdd(&B.counters[14], (u64)trips); atomicAdd(&B.counters[15], (u64)runGroups); }
#endif
}

// ------------------------------------------------------------------------------------------
// one wave per chain: stitch left extension + seed + right extension (extendWithOtherSeedChain /
// extendToFullSequenceLength, verboseSeedChain.cpp:23-136), then scoreOneAlignment (extensionAligner.cpp:52-182).
//
// Round 4.  The kernel waits on memory nine tenths of its time: a chain used to cost a dozen DEPENDENT round trips to cold rows (descriptors one after
// the other, the right extension moved, the seed copied, then the row just written read back for the likelihood, then lane 0 walking the levels for the
// first / last two).  Now: the descriptors of eight chains arrive together (one per lane, two round trips per eight chains); a lane owns CONSECUTIVE
// columns of the final row and fetches each from where it lives (pad: the read; left extension: in place; seed row; right extension at the end of the
// row) -- one round trip for every column of the chain --, writes what has to move, and keeps level / graph character / read character in registers for
// the likelihood terms and for the first / last levels (ballots, no loads).  The read qualities are the one dependent gather left.
template <int PER>
__device__ __forceinline__ void stitch_rounds(const DevBatch& B, const DevTables& T, const size_t cb, const int rOff, const int total, const int padL, const int nL, const int nSeed,
                                              const int nR, const int newEnd, const int stride, const int lane, double& llOut, int& f0, int& f1, int& l0, int& l1)
{
    const int seedAt = padL + nL, rightAt = seedAt + nSeed, padRAt = rightAt + nR;
    const int srcShift = (stride - nR) - rightAt;          // a column of the right extension sits this far behind its place (>= 0: the row holds the chain)
    double carry = 0.0; int baseIdx = 0;                   // running sum / read bases consumed before this round
    f0 = -1; f1 = -1; l0 = -1; l1 = -1;
    for(int c0 = 0; c0 < total; c0 += 64 * PER) {
        const int chunk = min(64 * PER, total - c0);
        const int per = (chunk + 63) / 64;
        const int j0 = c0 + lane * per, j1 = min(c0 + chunk, j0 + per);
        int lv[PER], ed[PER]; unsigned char gcv[PER], scv[PER], kind[PER];       // kind: 0 pad, 1 left extension (in place), 2 seed, 3 right extension, 4 none
        // ---- every column of the round from where it lives: all loads are requested before the first store.  (The right extension moves towards the
        // front of the row, destination <= source: a store of this round can only hit a source column of this or an earlier round, whose load precedes it
        // in program order -- and the pads behind it lie behind every source column still to be read.)
#pragma unroll
        for(int k = 0; k < PER; k++) {
            const int j = j0 + k; const bool in = k < per && j < j1;
            lv[k] = -1; ed[k] = -1; gcv[k] = '_'; scv[k] = '_'; kind[k] = 4;
            if(in) {
                if(j < padL) { kind[k] = 0; scv[k] = B.read_bases[rOff + j]; }
                else if(j < seedAt) { kind[k] = 1; lv[k] = B.ext_level[cb + j]; gcv[k] = B.ext_g[cb + j]; scv[k] = B.ext_s[cb + j]; }
                else if(j < rightAt) { const int q = j - seedAt; kind[k] = 2; lv[k] = B.seed_level[cb + q]; ed[k] = B.seed_edge[cb + q]; gcv[k] = B.seed_g[cb + q]; scv[k] = B.seed_s[cb + q]; }
                else if(j < padRAt) { const size_t p = cb + j + srcShift; kind[k] = 3; lv[k] = B.ext_level[p]; ed[k] = B.ext_edge[p]; gcv[k] = B.ext_g[p]; scv[k] = B.ext_s[p]; }
                else { kind[k] = 0; scv[k] = B.read_bases[rOff + newEnd + 1 + (j - padRAt)]; }
            }
        }
        WSYNC();
        // ---- read bases before each column -> the index of its quality; the gather of the qualities is requested, then the stores go out beside it
        int nb = 0;
#pragma unroll
        for(int k = 0; k < PER; k++) if(kind[k] != 4 && scv[k] != '_') nb++;
        int tot; const int before = baseIdx + wave_excl_scan(nb, tot);
        unsigned char qv[PER];
        {
            int idx = before;
#pragma unroll
            for(int k = 0; k < PER; k++) {
                qv[k] = 0;
                if(kind[k] != 4 && scv[k] != '_') { if(gcv[k] != '_') qv[k] = B.read_quals[rOff + idx]; idx++; }
            }
        }
#pragma unroll
        for(int k = 0; k < PER; k++) {
            const size_t o = cb + j0 + k;
            if(kind[k] == 1) B.ext_fromseed[o] = 0;
            else if(kind[k] != 4) {
                if(kind[k] != 3 || srcShift != 0) { B.ext_level[o] = lv[k]; B.ext_edge[o] = ed[k]; B.ext_g[o] = gcv[k]; B.ext_s[o] = scv[k]; }
                B.ext_fromseed[o] = kind[k] == 2 ? 1 : 0;
            }
        }
        // ---- scoreOneAlignment: terms are added strictly left to right in FP64 (same order as the reference loop).
        // The read quality of alignment-orientation base i is read_quals[i] in both strands: the reference indexes the
        // ORIGINAL-orientation read with len-i-1 when the chain is reverse (extensionAligner.cpp:85-88), which is the same base.
        // Every lane turns its columns into two addends each (an unused addend is +0.0, an exact identity); the running sum then walks the lanes in order.
        double t1[PER], t2[PER];
#pragma unroll
        for(int k = 0; k < PER; k++) {
            const unsigned char sc = scv[k], gc = gcv[k];
            double a1 = 0.0, a2 = 0.0;
            if(kind[k] != 4) {
                if(sc != '_') {
                    if(gc == '_') a1 = T.rate_ins_quarter;
                    else { a1 = T.rate_match_mismatch; a2 = (sc == gc) ? T.ll_match[qv[k]] : T.ll_mismatch[qv[k]]; }
                } else if(gc != '_') a1 = T.rate_indel;
            }
            t1[k] = a1; t2[k] = a2;
        }
        const int lanesUsed = (chunk + per - 1) / per;          // lanes beyond hold no column: nothing to add
        double acc = 0.0;
        for(int l = 0; l < lanesUsed; l++) {
            double in = __shfl(acc, l > 0 ? l - 1 : 0);
            if(lane == l) {
                double a = (l == 0) ? carry : in;
#pragma unroll
                for(int k = 0; k < PER; k++) { a += t1[k]; a += t2[k]; }
                acc = a;
            }
        }
        carry = __shfl(acc, lanesUsed - 1); baseIdx += tot;
        // ---- first / last two defined levels (verboseSeedChain.h:134-228) from the registers: per lane the number of defined levels among its columns,
        // the first two and the last two; lanes hold consecutive columns, so lane order is column order
        {
            int cnt = 0, a0 = -1, a1 = -1, z0 = -1, z1 = -1;
#pragma unroll
            for(int k = 0; k < PER; k++) if(kind[k] != 4 && lv[k] != -1) { if(cnt == 0) a0 = lv[k]; else if(cnt == 1) a1 = lv[k]; z1 = z0; z0 = lv[k]; cnt++; }
            const u64 m1 = __ballot(cnt >= 1);
            if(m1) {
                const int F = __ffsll((long long)m1) - 1, Lz = 63 - __clzll((long long)m1);
                if(f1 < 0) {
                    const int fa0 = __shfl(a0, F), fa1 = __shfl(a1, F);
                    int nx = -1; { const u64 m2 = m1 & ~(1ull << F); if(m2) nx = __shfl(a0, __ffsll((long long)m2) - 1); }
                    if(f0 < 0) { f0 = fa0; f1 = fa1 >= 0 ? fa1 : nx; }
                    else f1 = fa0;
                }
                const int lz0 = __shfl(z0, Lz), lz1 = __shfl(z1, Lz);
                int pv = -1; { const u64 m2 = m1 & ~(1ull << Lz); if(m2) pv = __shfl(z0, 63 - __clzll((long long)m2)); }
                const int second = lz1 >= 0 ? lz1 : (pv >= 0 ? pv : l0);        // (one defined level in this round: the second-last is the last of the rounds before)
                l0 = lz0; l1 = second;
            }
        }
    }
    llOut = carry;
}

// The chains of one draw: lane q holds chain cq (-1: none).  Descriptors of all of them in two round trips (the fields, then what they point to), then one
// chain after the other.  deferMode 1: the chains of deferred pairs stay pending (the second pass takes them), 0 / 2: every pending chain given.
__device__ __forceinline__ void stitch_draw(const DevBatch& B, const DevTables& T, const uint8_t* __restrict__ deferPairs, const int deferMode, const int lane, const int cq,
                                            u64& accChains, u64& accCols)
{
    const int stride = B.stride;
    const bool has = cq >= 0;
    int dSt = 0, dRead = 0, dNSeed = 0, dSB = 0, dSE = 0, dNcL = -1, dNcR = -1, dErrL = 0, dErrR = 0, dSbL = 0, dSeR = 0;
    if(has) dSt = B.ext_status[cq];
    if(has && dSt == EXT_PENDING) {
        dRead = B.chain_read[cq]; dNSeed = B.seed_ncols[cq]; dSB = B.seed_begin[cq]; dSE = B.seed_end[cq];
        const int2 nc = ((const int2*)B.dp_ncols)[cq], er = ((const int2*)B.dp_err)[cq];
        dNcL = nc.x; dNcR = nc.y; dErrL = er.x; dErrR = er.y; dSbL = B.dp_sb[2 * cq]; dSeR = B.dp_se[2 * cq + 1];
    }
    int dR0 = 0, dR1 = 0; bool dMine = has && dSt == EXT_PENDING;
    if(dMine) {
        dR0 = B.read_off[dRead]; dR1 = B.read_off[dRead + 1];
        // (fused entry point: the chains of a pair with a DP call in one of the side-stream classes stay pending in the first pass; the second
        // pass, which may run beside the first one, takes exactly those)
        if(deferMode == 1) dMine = deferPairs[dRead >> 1] == 0;
    }
    u64 mine = __ballot(dMine);
    for(; mine; mine &= mine - 1) {
        const int q = __ffsll((long long)mine) - 1;
        const int c = __builtin_amdgcn_readlane(cq, q);
        const int rOff = __builtin_amdgcn_readlane(dR0, q), seqLen = __builtin_amdgcn_readlane(dR1, q) - rOff;
        const size_t cb = row_base(B, c);
        const int nSeed = __builtin_amdgcn_readlane(dNSeed, q), sBegin = __builtin_amdgcn_readlane(dSB, q), sEnd = __builtin_amdgcn_readlane(dSE, q);
        const int ncL = __builtin_amdgcn_readlane(dNcL, q), ncR = __builtin_amdgcn_readlane(dNcR, q);
        const int errL = __builtin_amdgcn_readlane(dErrL, q), errR = __builtin_amdgcn_readlane(dErrR, q);
        int err = 0;
        if(errL || errR) err = ((errL <= -1000000) || (errR <= -1000000)) ? HLALA_CHAIN_ERR_COLUMNS : HLALA_CHAIN_ERR_FRONTIER;
        const bool haveL = ncL >= 0 && !errL, haveR = ncR >= 0 && !errR;
        const int nL = haveL ? ncL : 0, nR = haveR ? ncR : 0;
        const int newBegin = haveL ? __builtin_amdgcn_readlane(dSbL, q) : sBegin, newEnd = haveR ? __builtin_amdgcn_readlane(dSeR, q) : sEnd;
        const int padL = newBegin, padR = seqLen - 1 - newEnd;
        const int total = padL + nL + nSeed + nR + padR;
        if(!err && total > stride) err = HLALA_CHAIN_ERR_COLUMNS;
        if(err) {
            if(lane == 0) { B.ext_status[c] = err; B.ext_ncols[c] = 0; B.ext_ll[c] = (double)(errL ? errL : errR); atomicAdd(&B.counters[CNT_ERRORS], 1ull); }
        } else {
            double ll; int f0, f1, l0, l1;
            // chains of up to 192 columns (2x150 bp reads) take 3 columns per lane, longer ones 8 per lane in rounds of 512
            if(total <= 192) stitch_rounds<3>(B, T, cb, rOff, total, padL, nL, nSeed, nR, newEnd, stride, lane, ll, f0, f1, l0, l1);
            else stitch_rounds<8>(B, T, cb, rOff, total, padL, nL, nSeed, nR, newEnd, stride, lane, ll, f0, f1, l0, l1);
            if(lane == 0) {
                ((int4*)B.ext_firstlast)[c] = make_int4(f0, f1, l0, l1);
                B.ext_status[c] = HLALA_CHAIN_OK; B.ext_ncols[c] = total; B.ext_begin[c] = 0; B.ext_end[c] = seqLen - 1; B.ext_ll[c] = ll;
                accChains++; accCols += (u64)total;
            }
        }
        WSYNC();
    }
}

// deferMode 0: every pending chain; 1: all but the chains of deferred pairs (first pass of the fused entry point); 2: only those (second pass).
// Passes 0 / 1 draw `draw` consecutive chain NUMBERS per atomic (work_counter[7]), lane q of a draw looking at chain first + q: a third of the chains are pending, the
// others got their final status from k_dp_items.  Round 5, measured (profiles/r05_experiments.txt): with 8 chains per draw the kernel runs at the rate ONE L2 word
// hands out draws -- 762 k draws in 8.7 ms = 88 per microsecond (MI355X_MICROARCH.md: dequeue) -- but every way around the atomic cost more than it: draws of 64
// chains, or the dense position-ordered list of the pending chains, 15 ms; a static deal of 2 / 4 / 8 / 16 chains per wave and round 13 / 17 / 15 / 10 ms; draws of 12 or
// 16 chains are the optimum: 6.7 / 6.6 ms (the default: 12).  What
// the draws keep small is the WINDOW of 4 KB rows, out of 30 GB of column arrays, that the 5 120 waves touch at one time.
// Pass 2 does not look at chains at all: the waves take the PAIRS 64 at a time (no atomic), and the chains of the few deferred ones -- ~2.5 k pairs per million --
// are stitched; it used to test all 6.1 M chains beside the next batch's kernels (25 ms on the side stream for a few thousand chains; 0.15 ms now).
__global__ __launch_bounds__(64, 5) void k_stitch_chains(const DevGraph* __restrict__ Gp, const DevTables* __restrict__ Tp, const DevBatch* __restrict__ Bp,
                                                        const uint8_t* __restrict__ deferPairs, const int deferMode, const int draw, const int byRow)      // draw: chains per wave and round (1 .. 64)
{
    const DevBatch& B = *Bp;
    const DevTables& T = *Tp;
    const int lane = lane_id();
    u64 accChains = 0, accCols = 0;          // work counters, flushed once per wave (same-address atomics serialise at the L2)
    if(deferMode == 2) {
        const int nP = B.n_pairs;
        for(int p0 = (int)blockIdx.x * 64; p0 < nP; p0 += (int)gridDim.x * 64) {
            const int p = p0 + lane;
            u64 dm = __ballot(p < nP && deferPairs[p] != 0);
            for(; dm; dm &= dm - 1) {
                const int q = __ffsll((long long)dm) - 1;
                const int pp = p0 + q;
                const int c0 = uni(B.chain_off[2 * pp]), c1 = uni(B.chain_off[2 * pp + 2]);
                for(int cb = c0; cb < c1; cb += 64) stitch_draw(B, T, deferPairs, 0, lane, cb + lane < c1 ? cb + lane : -1, accChains, accCols);
            }
        }
    } else {
        // byRow: the draws walk the COLUMN ROWS (batch.h: chain_row -- only the chains that passed the filters hold one, in position order), not the chain numbers
        const bool rows = byRow && B.chain_order && B.chain_row;
        const int nList = rows ? ordered_chains(B) : B.n_chains;
        for(;;) {
            int w0 = 0;
            if(lane == 0) w0 = atomicAdd(&B.work_counter[7], draw);
            w0 = __builtin_amdgcn_readfirstlane(w0);
            if(w0 >= nList) break;
            const int wq = w0 + lane;
            int cq = (lane < draw && wq < nList) ? wq : -1;
            if(rows && cq >= 0) cq = B.chain_order[cq];
            stitch_draw(B, T, deferPairs, deferMode, lane, cq, accChains, accCols);
        }
    }
    if(lane == 0 && accChains) { atomicAdd(&B.counters[CNT_CHAINS_EXT], accChains); atomicAdd(&B.counters[CNT_OUT_COLS], accCols); }
}

}  // namespace hlala
