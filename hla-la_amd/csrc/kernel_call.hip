// kernel_call.hip -- the call of one locus from the all-pairs table (hla/HLATyper.cpp:2366-2541) on gfx950.
//
//   order        : pair indices by LL descending, Mism_avg ascending = two stable LSD radix sorts (hipCUB's device radix sort:
//                  a plain library sort; keys are the order-preserving integer images of the doubles)
//   P            : exp(LL - max) / sum; the sum is a fixed-shape tree over blocks (deterministic, not the reference's serial order:
//                  P agrees to ~1e-13 relative, the integer decisions below do not depend on it beyond that)
//   marginals    : the reference accumulates clusterI_overAllPairs while walking `order`; here every (cluster, rank, P) contribution is
//                  sorted by (cluster, rank) and one thread per cluster adds its contributions serially in rank order -- the same
//                  sequence of FP additions per cluster as the reference, so clusters with identical likelihoods keep identical sums
//   first/second : first maximum in cluster order (findIntMapMax), ties of the second allele by the smallest Mism_min.
#include <hipcub/hipcub.hpp>

#include "device_common.h"
#include "../../include/hlala_gpu.h"

namespace hlala {

// order-preserving map double -> u64 (ascending)
__device__ __forceinline__ u64 dbl_key(double d)
{
    u64 b = (u64)__double_as_longlong(d);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}

__device__ __forceinline__ long long tri_index(int c1, int c2, int C) { return (long long)c1 * C - (long long)c1 * (c1 - 1) / 2 + (c2 - c1); }

// per block: (max LL, first index achieving it); partials reduced by k_call_max_final
__global__ void k_call_max(const double* __restrict__ LL, long long n, double* __restrict__ pmax, long long* __restrict__ pidx)
{
    __shared__ double smax[256]; __shared__ long long sidx[256];
    double m = -INFINITY; long long mi = 0x7FFFFFFFFFFFFFFFll;
    for(long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        double v = LL[i];
        if(v > m || (v == m && i < mi)) { m = v; mi = i; }
    }
    smax[threadIdx.x] = m; sidx[threadIdx.x] = mi;
    __syncthreads();
    for(int s = 128; s > 0; s >>= 1) {
        if((int)threadIdx.x < s) {
            double v = smax[threadIdx.x + s]; long long vi = sidx[threadIdx.x + s];
            if(v > smax[threadIdx.x] || (v == smax[threadIdx.x] && vi < sidx[threadIdx.x])) { smax[threadIdx.x] = v; sidx[threadIdx.x] = vi; }
        }
        __syncthreads();
    }
    if(threadIdx.x == 0) { pmax[blockIdx.x] = smax[0]; pidx[blockIdx.x] = sidx[0]; }
}
__global__ void k_call_max_final(const double* __restrict__ pmax, const long long* __restrict__ pidx, int nb, double* __restrict__ outMax, long long* __restrict__ outIdx)
{
    double m = -INFINITY; long long mi = 0x7FFFFFFFFFFFFFFFll;
    for(int i = 0; i < nb; i++) if(pmax[i] > m || (pmax[i] == m && pidx[i] < mi)) { m = pmax[i]; mi = pidx[i]; }
    *outMax = m; *outIdx = mi;
}

__global__ void k_call_keys_mism(const double* __restrict__ MA, long long n, u64* __restrict__ key, int* __restrict__ idx)
{
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if(i < n) { key[i] = dbl_key(MA[i]); idx[i] = (int)i; }
}
// second pass: descending LL of the already mism-sorted indices
__global__ void k_call_keys_ll(const double* __restrict__ LL, const int* __restrict__ idx, long long n, u64* __restrict__ key)
{
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if(i < n) key[i] = ~dbl_key(LL[idx[i]]);
}

// P = exp(LL - max); per-block sums in a fixed tree
__global__ void k_call_p(const double* __restrict__ LL, long long n, const double* __restrict__ llMax, double* __restrict__ P, double* __restrict__ psum)
{
    __shared__ double ssum[256];
    const double mx = *llMax;
    double acc = 0;
    for(long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) { double p = exp(LL[i] - mx); P[i] = p; acc += p; }
    ssum[threadIdx.x] = acc;
    __syncthreads();
    for(int s = 128; s > 0; s >>= 1) { if((int)threadIdx.x < s) ssum[threadIdx.x] += ssum[threadIdx.x + s]; __syncthreads(); }
    if(threadIdx.x == 0) psum[blockIdx.x] = ssum[0];
}
__global__ void k_call_psum_final(const double* __restrict__ psum, int nb, double* __restrict__ out)
{
    double s = 0;
    for(int i = 0; i < nb; i++) s += psum[i];
    *out = s;
}
__global__ void k_call_normalize(double* __restrict__ P, long long n, const double* __restrict__ Psum)
{
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if(i >= n) return;
    const double s = *Psum;
    P[i] = (s > 0) ? P[i] / s : 1.0 / (double)n;                                              // :2421-2448
}

// pair index -> (c1, c2)
__global__ void k_call_clusters(int C, int* __restrict__ c1o, int* __restrict__ c2o)
{
    const int c1 = blockIdx.x;
    for(int c2 = c1 + threadIdx.x; c2 < C; c2 += blockDim.x) { long long i = tri_index(c1, c2, C); c1o[i] = c1; c2o[i] = c2; }
}

// contributions to clusterI_overAllPairs: (cluster << 32 | rank) -> P, two per pair (the second is a sentinel when c1 == c2), :2459-2486
__global__ void k_call_contrib(const int* __restrict__ order, const int* __restrict__ c1a, const int* __restrict__ c2a, const double* __restrict__ P, long long n,
                               u64* __restrict__ key, double* __restrict__ val, int* __restrict__ ties, const double* __restrict__ LL, const double* __restrict__ MA)
{
    long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if(r >= n) return;
    const int cI = order[r];
    const int c1 = c1a[cI], c2 = c2a[cI];
    const double p = P[cI];
    key[2 * r] = ((u64)(u32)c1 << 32) | (u64)(u32)r; val[2 * r] = p;
    key[2 * r + 1] = (c2 != c1) ? (((u64)(u32)c2 << 32) | (u64)(u32)r) : ~0ull; val[2 * r + 1] = p;
    if(r > 0) { const int pI = order[r - 1]; if(LL[pI] == LL[cI] && MA[pI] == MA[cI]) atomicAdd(ties, 1); }
}

// one thread per cluster: serial sum of its contributions in rank order
__global__ void k_call_marginals(const u64* __restrict__ key, const double* __restrict__ val, long long n2, int C, double* __restrict__ marg)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if(c >= C) return;
    const u64 lo = (u64)(u32)c << 32;
    long long a = 0, b = n2;                                   // first entry with key >= lo
    while(a < b) { long long m = (a + b) >> 1; if(key[m] < lo) a = m + 1; else b = m; }
    double s = 0;
    for(long long i = a; i < n2 && (key[i] >> 32) == (u64)(u32)c; i++) s += val[i];
    marg[c] = s;
}

// first / second allele, :2490-2533 (single block)
__global__ void k_call_decide(int C, const double* __restrict__ marg, const double* __restrict__ P, const double* __restrict__ MM, const double* __restrict__ llMax,
                              const long long* __restrict__ llIdx, const int* __restrict__ ties, hlala_call_out* __restrict__ out)
{
    if(threadIdx.x != 0 || blockIdx.x != 0) return;
    double max = 0; int first = 0;
    for(int c = 0; c < C; c++) if(c == 0 || marg[c] > max) { max = marg[c]; first = c; }                 // findIntMapMax: first maximum in key order
    double bestP = 0; bool have = false;
    for(int x = 0; x < C; x++) { const int a = x < first ? x : first, b = x < first ? first : x; const double p = P[tri_index(a, b, C)]; if(!have || p > bestP) { bestP = p; have = true; } }
    double bestM = 0; int second = 0; have = false;
    for(int x = 0; x < C; x++) {
        const int a = x < first ? x : first, b = x < first ? first : x; const long long i = tri_index(a, b, C);
        if(P[i] == bestP) { const double v = -1 * MM[i]; if(!have || v > bestM) { bestM = v; second = x; have = true; } }
    }
    out->first_cluster = first; out->second_cluster = second; out->first_marginal = max; out->second_p = bestP;
    out->ll_max = *llMax; out->max_pair = (int)*llIdx; out->n_sort_ties = *ties;
}

}  // namespace hlala
