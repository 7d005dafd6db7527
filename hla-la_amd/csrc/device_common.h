// device_common.h -- device-side data layout shared by the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace hlala {

typedef unsigned long long u64;
typedef unsigned int u32;

// Flattened PRG resident in HBM (see flat_graph.hpp for the meaning of each array).
struct DevGraph {
    int L, N, E, P;
    const int* level_off;      // [L+1]
    const int* node_level;     // [N]
    const int* node_orig;      // [N]
    const int* out_off;        // [N+1]
    const int* out_to;         // [E]
    const uint8_t* out_label;  // [E]
    const int* out_eid;        // [E]
    const int* in_off;
    const int* in_from;
    const uint8_t* in_label;
    const int* in_eid;
    const unsigned int* in_rec; // [E] packed in-edge records of the projection's re-threading DP (flat_graph.hpp)
    const uint8_t* level_fast;  // [L] the level can be solved edge-parallel (flat_graph.hpp)
    const int* edge_from_new;  // [E] by creation index
    const int* edge_to_new;    // [E]
    const uint8_t* edge_label; // [E] by creation index
    const int* jf_off; const int* jf_node; const int* jf_path; const int* jf_lvl;
    const int* jb_off; const int* jb_node; const int* jb_path; const int* jb_lvl;
    const uint8_t* out_prank; const uint8_t* in_prank; const uint8_t* jf_prank; const uint8_t* jb_prank;   // rank among the node's earlier entries to the same target (flat_graph.hpp)
    const uint8_t* jfree_out; const uint8_t* jfree_in;   // [L] levels without a gap-path jump from this level on, in either direction (flat_graph.hpp)
    const u32* lin_label; const uint8_t* lin_out; const uint8_t* lin_in; const int* lin_eid;   // [L] linear steps and their run lengths (flat_graph.hpp; kernel_dp_band.hip)
    const u64* trk_w_out; const u64* trk_w_in; const uint8_t* trk_out; const uint8_t* trk_in; const u32* trk_j_out; const u32* trk_j_in; const int* trk_jp_out; const int* trk_jp_in;   // [L] track steps (flat_graph.hpp; kernel_dp_band2.hip)
    const int4* nrec_out;      // [2*N] 32-byte node records of the extension DP (flat_graph.hpp)
    const int4* nrec_in;
    const int* path_len;       // [P]
    const long long* path_off; // [P+1]
    const int* path_edges;
    const uint8_t* gap_stretch;// [L-1]
    const long long* lp_off;   // [L+1]
    const int* lp_seqid;
    const int* lp_pos;
};

// Constant tables computed once on the host with the host libm so that device results are
// bit-identical to a CPU evaluation of the reference formulas.
struct DevTables {
    double ll_match[256];      // log(pCorrect(q)) with the 0.999 cap / 1e-5 floor, extensionAligner.cpp:124-141
    double ll_mismatch[256];   // log((1 - pCorrect(q)) / 3)
    double rate_indel;         // log(0.001) or log(0.075)
    double rate_ins_quarter;   // rate_insertions + log(1/4)
    double rate_match_mismatch;// log(1 - 2 * rate)
    double is_penalty;         // log pdf(mean + 8 sd), processBAM.cpp:2343-2346
    int    is_dmin, is_n;      // logpdf table covers integer distances [is_dmin, is_dmin + is_n)
    const double* is_logpdf;   // device pointer; entries with pdf <= 0 hold is_penalty
    double phred_thr[256];     // phred_thr[k] = largest pWrong for which PCorrectToPhred gives >= k
    double pcorrect[256];      // Utilities::PhredToPCorrect(q), Utilities.cpp:357-377 (host libm)
};

// capacities of the extension DP (one wavefront per chain)
constexpr int DP_WCAP      = 128;    // frontier cells per diagonal list
constexpr int DP_HC        = 512;    // candidate-target hash entries per iteration
constexpr int DP_SEQCAP    = 1024;   // read length staged in LDS
constexpr int DP_CELLS     = 16384;  // kept cells per DP call (global slab); < 32768 so slots fit a short
constexpr int DP_EARLY     = 4096;   // hash entries for cells reached early through gap-path jumps
constexpr int DP_STEPS     = 8192;   // backtrace steps
constexpr int DP_COMPLETED = 2048;   // sequence-complete cells
constexpr int DP_CELLS_LARGE     = 65536;  // kept cells per DP call of the large-capacity class (<= 131072: 17-bit slots in the back pointers)
constexpr int DP_COMPLETED_LARGE = 16384;  // sequence-complete cells, large-capacity class
constexpr int DP_MAX_PARALLEL = 127; // parallel edges (or gap paths) between ONE pair of nodes: what the 7-bit rank in the push index of a DP candidate holds (kernel_dp.hip); a node's degree is not limited
constexpr int DP_NEG       = -30000; // minusInfinity (-DBL_MAX in the reference, extensionAligner.cpp:363)

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }
// A pointer that is KNOWN to point into HBM.  A pointer the compiler cannot trace back to a kernel argument -- one read from a descriptor struct or from LDS -- is
// 'generic': every access through it is a FLAT instruction, which goes to the LDS queue as well as to the vector-memory path, counts in both wait counters and can
// only be waited for with 'everything outstanding' (round 6: 286 of the 310 loads of the long-read projection were flat_load).  glob() casts to the global address
// space (the result is a pointer TYPE of that address space: `auto`, not `T*`, receives it); GPtr<T> is a pointer member whose accesses are global.
#define HLALA_AS_GLOBAL __attribute__((address_space(1)))
template <class T> struct GPtr {
    typedef HLALA_AS_GLOBAL T GT;
    T* p;
    __device__ __forceinline__ GPtr& operator=(T* q) { p = q; return *this; }
    __device__ __forceinline__ GT* g() const { return (GT*)p; }
    __device__ __forceinline__ GT& operator[](int i) const { return g()[i]; }
    __device__ __forceinline__ GT& operator[](unsigned i) const { return g()[i]; }
    __device__ __forceinline__ GT& operator[](long long i) const { return g()[i]; }
    __device__ __forceinline__ GT& operator[](size_t i) const { return g()[i]; }
    __device__ __forceinline__ operator GT*() const { return g(); }
};
template <class T> __device__ __forceinline__ HLALA_AS_GLOBAL T* glob(T* p) { return (HLALA_AS_GLOBAL T*)p; }
// All kernels run ONE wavefront per block, so a block barrier is only a memory-ordering point between lanes of the
// same wave.  __syncthreads() is not used: hipcc (ROCm 7.2) miscompiled persistent work loops that `continue` / `break`
// around it (kernels never terminated); a wavefront-scope fence + scheduling barrier gives the ordering without the
// workgroup-barrier semantics.
#define WSYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while(0)
// make a value the compiler cannot prove wave-uniform explicitly scalar (all 64 lanes hold the same value)
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
// dynamic work distribution for one-wave blocks: lane 0 draws the next item, every lane gets it in an SGPR
__device__ __forceinline__ int next_work(int* counter)
{
    int w = 0;
    if(lane_id() == 0) w = atomicAdd(counter, 1);
    return __builtin_amdgcn_readfirstlane(w);
}

// Wave-wide reductions on the DPP cross-lane path (row_shr 1/2/3 + row_bcast15/31): ~6 VALU ops instead of six
// ds_bpermute round trips through the LDS crossbar.  The result is read from lane 63 into an SGPR.
template <int CTRL>
__device__ __forceinline__ int dpp_mov(int oldv, int v) { return __builtin_amdgcn_update_dpp(oldv, v, CTRL, 0xF, 0xF, false); }
#define HLALA_DPP_REDUCE(v, IDENT, OP)                                                     \
    do {                                                                                    \
        int t_;                                                                             \
        t_ = dpp_mov<0x111>(IDENT, v); v = OP(v, t_);   /* row_shr:1 */                     \
        t_ = dpp_mov<0x112>(IDENT, v); v = OP(v, t_);   /* row_shr:2 */                     \
        t_ = dpp_mov<0x114>(IDENT, v); v = OP(v, t_);   /* row_shr:4 */                     \
        t_ = dpp_mov<0x118>(IDENT, v); v = OP(v, t_);   /* row_shr:8 */                     \
        t_ = __builtin_amdgcn_update_dpp(IDENT, v, 0x142, 0xA, 0xF, false); v = OP(v, t_);  /* row_bcast:15 */ \
        t_ = __builtin_amdgcn_update_dpp(IDENT, v, 0x143, 0xC, 0xF, false); v = OP(v, t_);  /* row_bcast:31 */ \
    } while(0)
__device__ __forceinline__ int op_max_(int a, int b) { return a > b ? a : b; }
__device__ __forceinline__ int op_add_(int a, int b) { return a + b; }
__device__ __forceinline__ int wave_max_i32(int v)
{
    HLALA_DPP_REDUCE(v, (int)0x80000000, op_max_);
    return __builtin_amdgcn_readlane(v, 63);
}
__device__ __forceinline__ int wave_sum_i32(int v)
{
    HLALA_DPP_REDUCE(v, 0, op_add_);
    return __builtin_amdgcn_readlane(v, 63);
}
__device__ __forceinline__ u64 wave_min_u64(u64 v)
{
    // two 32-bit max passes on the complemented halves: high word first, then the low word among the winners
    int hi = (int)(~(u32)(v >> 32) ^ 0x80000000u);
    int mh = wave_max_i32(hi);
    int lo = (hi == mh) ? (int)(~(u32)v ^ 0x80000000u) : (int)0x80000000;
    int ml = wave_max_i32(lo);
    return ((u64)(~((u32)mh ^ 0x80000000u)) << 32) | (u64)(~((u32)ml ^ 0x80000000u));
}
// exclusive prefix sum over the wave; returns the lane's offset, `total` is wave-uniform
// exp(x) for x <= 0, correctly rounded (double-double evaluation: ln2 in three exact pieces, 24 Taylor terms with double-double
// coefficients 1/k!).  The posteriors of the pairing step decide integer outputs at the p == 1 boundary (PCorrectToPhred, Utilities.cpp:178-203):
// the host's libm returns the correctly rounded exponential in all but rare cases, the device library's exp is off by an ulp in one call of
// ten.  Arguments below -700 (results near the subnormal range, irrelevant to a normalised posterior) take the library exp.
__device__ inline double exp_cr_nonpos(double x)
{
#pragma clang fp contract(off)      // the error-free transformations below must not be fused behind their back
    if(!(x < 0.0)) return 1.0;
    if(x < -700.0) return exp(x);
    const double n = rint(x * 0x1.71547652b82fep+0);
    // r = x - n ln2, exact in the first two steps (n has 11 bits, the pieces 32)
    const double r0 = fma(-n, 0x1.62e42fee00000p-1, x);
    const double t1 = n * 0x1.a39ef35600000p-33;                    // exact
    double rh = r0 - t1; double bb = rh - r0; double rl = (r0 - (rh - bb)) + (-t1 - bb);
    const double t2 = n * 0x1.93c7673007e5fp-65;
    { double s = rh - t2; double b2 = s - rh; double e = (rh - (s - b2)) + (-t2 - b2); e += rl; rh = s + e; rl = e - (rh - s); }
    const double ch[25] = {0x1.0000000000000p+0, 0x1.0000000000000p+0, 0x1.0000000000000p-1, 0x1.5555555555555p-3, 0x1.5555555555555p-5, 0x1.1111111111111p-7, 0x1.6c16c16c16c17p-10,
        0x1.a01a01a01a01ap-13, 0x1.a01a01a01a01ap-16, 0x1.71de3a556c734p-19, 0x1.27e4fb7789f5cp-22, 0x1.ae64567f544e4p-26, 0x1.1eed8eff8d898p-29, 0x1.6124613a86d09p-33,
        0x1.93974a8c07c9dp-37, 0x1.ae7f3e733b81fp-41, 0x1.ae7f3e733b81fp-45, 0x1.952c77030ad4ap-49, 0x1.6827863b97d97p-53, 0x1.2f49b46814157p-57, 0x1.e542ba4020225p-62,
        0x1.71b8ef6dcf572p-66, 0x1.0ce396db7f853p-70, 0x1.761b41316381ap-75, 0x1.f2cf01972f578p-80};
    const double cl[25] = {0.0, 0.0, 0.0, 0x1.5555555555555p-57, 0x1.5555555555555p-59, 0x1.1111111111111p-63, -0x1.f49f49f49f49fp-65, 0x1.a01a01a01a01ap-73, 0x1.a01a01a01a01ap-76,
        -0x1.c154f8ddc6c00p-73, 0x1.cbbc05b4fa99ap-76, -0x1.c062e06d1f209p-80, -0x1.2aec959e14c06p-83, 0x1.f28e0cc748ebep-87, 0x1.05d6f8a2efd1fp-92, 0x1.1d8656b0ee8cbp-97,
        0x1.1d8656b0ee8cbp-101, 0x1.ac981465ddc6cp-103, 0x1.eec01221a8b0bp-107, 0x1.2650f61dbdcb4p-112, 0x1.ea72b4afe3c2fp-120, -0x1.d043ae40c4647p-120, -0x1.aebcdbd20331cp-124,
        -0x1.3423c7d91404fp-130, -0x1.9ada5fcc1ab14p-135};
    double ph = ch[24], pl = cl[24];
#pragma unroll
    for(int k = 23; k >= 0; k--) {
        // (ph, pl) = (ph, pl) * (rh, rl) + (ch[k], cl[k])
        const double m = ph * rh; double e = fma(ph, rh, -m); e = fma(ph, rl, e); e = fma(pl, rh, e);
        double mh = m + e; double ml = e - (mh - m);
        const double s = mh + ch[k]; const double b2 = s - mh; double e2 = (mh - (s - b2)) + (ch[k] - b2); e2 += ml + cl[k];
        ph = s + e2; pl = e2 - (ph - s);
    }
    return ldexp(ph, (int)n);
}

__device__ __forceinline__ unsigned long long wave_sum_u64(unsigned long long v)
{
    for(int o = 32; o > 0; o >>= 1) {
        const unsigned lo = (unsigned)__shfl_xor((int)(unsigned)v, o), hi = (unsigned)__shfl_xor((int)(unsigned)(v >> 32), o);
        v += ((unsigned long long)hi << 32) | lo;
    }
    return v;
}

// inclusive scans over the wave: the DPP sequence above leaves in every lane the reduction of lanes 0 .. lane (row_shr within a row of 16,
// then the last lane of rows 0 / 2 into rows 1 / 3, then lane 31 into rows 2 and 3)
__device__ __forceinline__ int wave_incl_scan_add(int v) { HLALA_DPP_REDUCE(v, 0, op_add_); return v; }
__device__ __forceinline__ int wave_incl_scan_max(int v) { HLALA_DPP_REDUCE(v, (int)0x80000000, op_max_); return v; }
__device__ __forceinline__ int wave_excl_scan(int v, int& total)
{
    const int x = wave_incl_scan_add(v);
    total = __builtin_amdgcn_readlane(x, 63);
    return x - v;
}

// glibc rand_r (stdlib/rand_r.c, glibc 2.35: three rounds of the 1103515245/12345 LCG yielding
// 11 + 10 + 10 bits) as called by Utilities::randomNumber_nonCritical (Utilities.cpp:922-927).
__host__ __device__ inline int glibc_rand_r(unsigned int* seed)
{
    unsigned int next = *seed;
    int result;
    next *= 1103515245u; next += 12345u;
    result = (unsigned int)(next / 65536u) % 2048u;
    next *= 1103515245u; next += 12345u;
    result <<= 10; result ^= (unsigned int)(next / 65536u) % 1024u;
    next *= 1103515245u; next += 12345u;
    result <<= 10; result ^= (unsigned int)(next / 65536u) % 1024u;
    *seed = next;
    return result;
}

// Utilities::PCorrectToPhred (Utilities.cpp:178-203) through the host-computed threshold table.
__device__ inline unsigned char phred_from_pcorrect(const DevTables& T, double pCorrect)
{
    double pWrong = 1 - pCorrect;
    if(pWrong == 0) pWrong = 1e-100;
    // result k = largest k with pWrong <= phred_thr[k]; thresholds are non-increasing in k
    int lo = 0, hi = 255;
    while(lo < hi) { int mid = (lo + hi + 1) >> 1; if(pWrong <= T.phred_thr[mid]) lo = mid; else hi = mid - 1; }
    return (unsigned char)lo;
}

}  // namespace hlala
