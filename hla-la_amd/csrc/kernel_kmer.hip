// kernel_kmer.hip -- which k-mers of the called alleles occur in the reads (proportionkMersCovered, hla/HLATyper.cpp:999-1027 and
// :2652-2688) on gfx950.
//
// The reference hashes every k-mer of every read into an unordered_map<string,int> and then asks it a few hundred questions per
// locus.  Here nothing is indexed: the questions (canonical 2-bit codes of the query k-mers, sorted, in LDS) stay put and every
// k-mer of every read is looked up in them -- one wavefront per read, a tile of the read staged in LDS as 2-bit codes, one lane
// per k-mer start, a binary search over <= 4096 sorted u64.  HBM traffic = the read bases once.
#include "device_common.h"
#include "batch.h"

namespace hlala {

constexpr int KMER_TILE = 256;          // k-mer start positions per tile
constexpr int KMER_QCAP = 4096;         // queries held in LDS

__device__ __forceinline__ int base_code(unsigned char c) { return c == 'A' ? 0 : c == 'C' ? 1 : c == 'G' ? 2 : c == 'T' ? 3 : 4; }

// canonical form = the lexicographically smaller of the k-mer and its reverse complement (kMer_canonical_representation,
// hla/HLATyper.cpp:4237-4256); with A<C<G<T -> 0..3, first base in the top bits, that is the smaller integer
// One read, one wavefront: every k-mer of bases[0 .. len) is looked up in the sorted queries q[0 .. nQ) (LDS).
__device__ __forceinline__ void kmer_scan_read(const uint8_t* __restrict__ bases, int len, int k, int nQ, const u64* q, unsigned char* code, uint8_t* __restrict__ present, int lane)
{
    for(int t0 = 0; t0 + k <= len; t0 += KMER_TILE) {
        const int nb = min(KMER_TILE + k - 1, len - t0);                   // bases of this tile
        __syncthreads();
        for(int i = lane; i < nb; i += 64) code[i] = (unsigned char)base_code(bases[t0 + i]);
        __syncthreads();
        const int nStarts = nb - k + 1;
        for(int s = lane; s < nStarts; s += 64) {
            u64 f = 0, rc = 0; bool ok = true;
            for(int j = 0; j < k; j++) {
                const int cj = code[s + j];
                ok = ok && cj < 4;
                f = (f << 2) | (u64)(cj & 3);
                rc |= (u64)(3 - (cj & 3)) << (2 * j);
            }
            if(!ok) continue;
            const u64 canon = rc < f ? rc : f;
            int a = 0, b = nQ;
            while(a < b) { const int m = (a + b) >> 1; if(q[m] < canon) a = m + 1; else b = m; }
            if(a < nQ && q[a] == canon) present[a] = 1;
        }
    }
}

__global__ __launch_bounds__(64) void k_kmer_presence(const DevBatch* __restrict__ Bp, const uint8_t* __restrict__ pair_mask, int k, int nQ,
                                                      const u64* __restrict__ queries, uint8_t* __restrict__ present)
{
    const DevBatch& B = *Bp;
    __shared__ u64 q[KMER_QCAP];
    __shared__ unsigned char code[KMER_TILE + 32];
    const int lane = threadIdx.x;
    for(int i = lane; i < nQ; i += 64) q[i] = queries[i];
    __syncthreads();
    const int nReads = B.unpaired ? B.n_pairs : 2 * B.n_pairs;
    for(int r = blockIdx.x; r < nReads; r += gridDim.x) {
        if(pair_mask && !pair_mask[B.unpaired ? r : r / 2]) continue;
        const int r0 = B.read_off[r];
        kmer_scan_read(B.read_bases + r0, B.read_off[r + 1] - r0, k, nQ, q, code, present, lane);
    }
}

// ---- the reads the typing looks at, kept on the device batch by batch while the batches are resident (hlala_kmer_keep_reads): the k-mer questions come
//      after the calls, when most batches of a sample have long been released -- asking them of the kept reads saves uploading every batch a second time.
// tot[0] += reads, tot[1] += bases of the looked-at units of the batch
__global__ __launch_bounds__(256) void k_kmer_count_kept(const DevBatch* __restrict__ Bp, const uint8_t* __restrict__ pair_mask, unsigned long long* __restrict__ tot)
{
    const DevBatch& B = *Bp;
    const int nReads = B.unpaired ? B.n_pairs : 2 * B.n_pairs;
    unsigned long long n = 0, bytes = 0;
    for(int r = blockIdx.x * blockDim.x + threadIdx.x; r < nReads; r += gridDim.x * blockDim.x)
        if(!pair_mask || pair_mask[B.unpaired ? r : r / 2]) { n++; bytes += (unsigned long long)(B.read_off[r + 1] - B.read_off[r]); }
    for(int o = 32; o > 0; o >>= 1) { n += __shfl_down(n, o); bytes += __shfl_down(bytes, o); }
    if((threadIdx.x & 63) == 0 && n) { atomicAdd(&tot[0], n); atomicAdd(&tot[1], bytes); }
}
// a wavefront per looked-at read: a place in the store (any order: presence is a union over the reads), then the copy
__global__ __launch_bounds__(64) void k_kmer_keep(const DevBatch* __restrict__ Bp, const uint8_t* __restrict__ pair_mask, unsigned long long* __restrict__ cursor,
                                                  long long* __restrict__ start, int* __restrict__ length, uint8_t* __restrict__ store)
{
    const DevBatch& B = *Bp;
    const int lane = threadIdx.x;
    const int nReads = B.unpaired ? B.n_pairs : 2 * B.n_pairs;
    for(int r = blockIdx.x; r < nReads; r += gridDim.x) {
        if(pair_mask && !pair_mask[B.unpaired ? r : r / 2]) continue;
        const int r0 = B.read_off[r], len = B.read_off[r + 1] - r0;
        unsigned long long at = 0;
        if(lane == 0) { const unsigned long long slot = atomicAdd(&cursor[0], 1ull); at = atomicAdd(&cursor[1], (unsigned long long)len); start[slot] = (long long)at; length[slot] = len; }
        at = __shfl(at, 0);
        for(int i = lane; i < len; i += 64) store[at + (unsigned long long)i] = B.read_bases[r0 + i];
    }
}
__global__ __launch_bounds__(64) void k_kmer_presence_kept(const long long* __restrict__ start, const int* __restrict__ length, const uint8_t* __restrict__ store, int nReads,
                                                           int k, int nQ, const u64* __restrict__ queries, uint8_t* __restrict__ present)
{
    __shared__ u64 q[KMER_QCAP];
    __shared__ unsigned char code[KMER_TILE + 32];
    const int lane = threadIdx.x;
    for(int i = lane; i < nQ; i += 64) q[i] = queries[i];
    __syncthreads();
    for(int r = blockIdx.x; r < nReads; r += gridDim.x) kmer_scan_read(store + start[r], length[r], k, nQ, q, code, present, lane);
}

}  // namespace hlala
