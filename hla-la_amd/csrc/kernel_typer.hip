// kernel_typer.hip -- HLATyper per-read scoring (kernels D and E of SURVEY.md 2.1).
//   k_exon_loglik : likelihoods_perCluster_perRead / mismatches_perCluster_perRead   hla/HLATyper.cpp:2067-2277
//   k_pair_loglik : all cluster pairs, sum over reads of logAvg                       hla/HLATyper.cpp:2293-2364, Utilities.cpp:1368-1379
// D is table-driven (host libm) and adds the per-position terms in the reference's order: bit-identical to the CPU.
// E evaluates exp / log on the device in FP64 (transcendental-bound, not HBM-bound): parity within 1e-9 relative.
#include "device_common.h"

namespace hlala {

struct TyperTables {
    double ll_match[256];            // log(pCorrect): cap 0.999 (veryConservativeReadLikelihoods), pCorrect == 0 -> 0.001 (:2188-2200)
    double ll_mismatch[256];         // log((1 - pCorrect) * (1/3))
    double ll_ins_actual;            // log(insertionP) + log(1/4)   (:952-954)
    double ll_deletion;              // log(deletionP)
    double ll_match_mismatch;        // log(1 - insertionP - deletionP)
};

// one thread per (cluster, read); consecutive threads = consecutive clusters, cluster sequences transposed to [P][C]
__global__ void k_exon_loglik(const TyperTables* __restrict__ Tp, int C, int P, int R, const uint8_t* __restrict__ seqT,
                              const int* __restrict__ pos_off, const int* __restrict__ pos_exon, const uint8_t* __restrict__ pos_g0,
                              const int* __restrict__ pos_glen, const uint8_t* __restrict__ pos_qual, const uint8_t* __restrict__ pos_use,
                              double* __restrict__ LL, int* __restrict__ mism)
{
    const TyperTables& T = *Tp;
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    int r = blockIdx.y;
    if(c >= C || r >= R) return;
    double log_likelihood_read = 0; int mismatches = 0;
    for(int i = pos_off[r]; i < pos_off[r + 1]; i++) {
        if(!pos_use[i]) continue;
        unsigned char e = seqT[(size_t)pos_exon[i] * C + c];
        unsigned char g0 = pos_g0[i]; int glen = pos_glen[i];
        int l_diff = glen - 1;
        double lp = 0;
        if(e == '_') {
            if(!(glen == 1 && g0 == '_')) lp += (T.ll_ins_actual * (1 + l_diff));            // :2163
        } else {
            if(g0 == '_') lp += T.ll_deletion;                                                // :2177
            else {
                lp += T.ll_match_mismatch;                                                    // :2186
                unsigned char q = pos_qual[i];
                lp += (e == g0) ? T.ll_match[q] : T.ll_mismatch[q];
            }
            lp += (T.ll_ins_actual * l_diff);                                                 // :2225
        }
        if(!(glen == 1 && g0 == '_')) if(!(glen == 1 && g0 == e)) mismatches++;             // :2236-2242
        log_likelihood_read += lp;
    }
    LL[(size_t)c * R + r] = log_likelihood_read;
    mism[(size_t)c * R + r] = mismatches;
}

__device__ __forceinline__ double log_avg(double a, double b)                                 // Utilities::logAvg, Utilities.cpp:1368-1379
{
    if(a > b) return (log(0.5) + (log(1 + exp(b - a)) + a));
    return (log(0.5) + (log(1 + exp(a - b)) + b));
}

// block = PAIRLL_ROWS consecutive c1 and 256 consecutive c2 >= the first of them: the rows are staged in LDS, the operand of column c2 for
// read r (transposed matrix: coalesced) is loaded ONCE and meets all the rows -- a quarter of the operand traffic of one row per block, and
// four independent exp / log chains per thread to fill the FP64 pipeline.  Every (c1, c2) sum still runs over the reads left to right, as the
// reference's loop does (:2312-2345).  Cells below the diagonal (c2 < c1) of the first rows' tile are computed and dropped.
#ifndef HLALA_PAIRLL_ROWS
#define HLALA_PAIRLL_ROWS 4
#endif
constexpr int PAIRLL_TILE = 512, PAIRLL_ROWS = HLALA_PAIRLL_ROWS;
__global__ __launch_bounds__(256) void k_pair_loglik(int C, int R, const double* __restrict__ LL, const double* __restrict__ LLT,
                                                     const int* __restrict__ mism, const int* __restrict__ mismT,
                                                     double* __restrict__ pairLL, double* __restrict__ misAvg, double* __restrict__ misMin)
{
    __shared__ double rowA[PAIRLL_ROWS][PAIRLL_TILE];
    __shared__ int rowM[PAIRLL_ROWS][PAIRLL_TILE];
    const int c1b = blockIdx.y * PAIRLL_ROWS;
    const int c2 = c1b + blockIdx.x * blockDim.x + threadIdx.x;
    if(c1b + (int)(blockIdx.x * blockDim.x) >= C) return;
    double ll[PAIRLL_ROWS], sAvg[PAIRLL_ROWS], sMin[PAIRLL_ROWS];
#pragma unroll
    for(int k = 0; k < PAIRLL_ROWS; k++) { ll[k] = 0; sAvg[k] = 0; sMin[k] = 0; }
    for(int r0 = 0; r0 < R; r0 += PAIRLL_TILE) {
        const int n = min(PAIRLL_TILE, R - r0);
        __syncthreads();
        for(int k = 0; k < PAIRLL_ROWS; k++) {
            const int c1 = min(c1b + k, C - 1);
            for(int i = threadIdx.x; i < n; i += blockDim.x) { rowA[k][i] = LL[(size_t)c1 * R + r0 + i]; rowM[k][i] = mism[(size_t)c1 * R + r0 + i]; }
        }
        __syncthreads();
        if(c2 < C)
            for(int i = 0; i < n; i++) {
                const double b = LLT[(size_t)(r0 + i) * C + c2];
                const int m2 = mismT[(size_t)(r0 + i) * C + c2];
#pragma unroll
                for(int k = 0; k < PAIRLL_ROWS; k++) {
                    const double a = rowA[k][i]; const int m1 = rowM[k][i];
                    ll[k] += log_avg(a, b);
                    sAvg[k] += ((double)(m1 + m2) / 2.0);
                    sMin[k] += (m1 < m2) ? m1 : m2;
                }
            }
    }
#pragma unroll
    for(int k = 0; k < PAIRLL_ROWS; k++) {
        const int c1 = c1b + k;
        if(c1 < C && c2 >= c1 && c2 < C) {
            const size_t idx = (size_t)c1 * C - (size_t)c1 * (c1 - 1) / 2 + (size_t)(c2 - c1);
            pairLL[idx] = ll[k]; misAvg[idx] = sAvg[k]; misMin[idx] = sMin[k];
        }
    }
}

template <class T>
__global__ void k_transpose(int rows, int cols, const T* __restrict__ in, T* __restrict__ out)
{
    __shared__ T tile[32][33];
    int x = blockIdx.x * 32 + threadIdx.x, y = blockIdx.y * 32 + threadIdx.y;
    if(x < cols && y < rows) tile[threadIdx.y][threadIdx.x] = in[(size_t)y * cols + x];
    __syncthreads();
    int tx = blockIdx.y * 32 + threadIdx.x, ty = blockIdx.x * 32 + threadIdx.y;
    if(tx < rows && ty < cols) out[(size_t)ty * rows + tx] = tile[threadIdx.x][threadIdx.y];
}

}  // namespace hlala
