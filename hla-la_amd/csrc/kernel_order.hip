// kernel_order.hip -- position order of a batch's chains.
//
// A batch arrives in read-name order (the reference's std::map of read IDs, mapper/processBAM.cpp:2024-2039), i.e. in RANDOM order of graph
// position: the persistent kernels of stages A and B, which hand out chains / DP calls from a list, then have a few thousand waves on a few
// thousand unrelated places of a 0.7 GB graph, and every node record, CSR window and level list they touch comes from HBM (measured, round 3:
// 43 GB fetched by the 16-lane DP kernel of a 1 M-pair batch against 0.44 GB of node records in the whole graph).  Results do not depend on the order
// in which chains are processed (every DP call draws its random seed from its chain's absolute number), so the lists are put into position order:
// a counting sort of the chains that passed the filters by (first level >> shift) -- buckets of a few hundred levels; the order inside a bucket is left to
// the atomics.
// The work in flight at one time then covers a window of the graph that fits the 4 MB L2 of an XCD.
//   k_filter_chains (kernel_project.hip)  bucket of every chain + histogram
//   k_order_scan                           exclusive scan of the histogram (one block)
//   k_order_scatter                        chain numbers into their bucket's range
#include "batch.h"

namespace hlala {

constexpr int ORDER_SCAN_THREADS = 1024;

__global__ __launch_bounds__(ORDER_SCAN_THREADS) void k_order_scan(int* hist, int nb)
{
    __shared__ int part[ORDER_SCAN_THREADS];
    const int t = threadIdx.x;
    const int per = (nb + ORDER_SCAN_THREADS - 1) / ORDER_SCAN_THREADS;
    const int a = t * per, z = min(nb, a + per);
    int s = 0;
    for(int i = a; i < z; i++) s += hist[i];
    part[t] = s;
    __syncthreads();
    for(int o = 1; o < ORDER_SCAN_THREADS; o <<= 1) {
        const int v = t >= o ? part[t - o] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    int run = part[t] - s;          // exclusive start of this thread's stretch
    for(int i = a; i < z; i++) { const int h = hist[i]; hist[i] = run; run += h; }
}

__global__ void k_order_scatter(const DevBatch* __restrict__ Bp)
{
    const DevBatch& B = *Bp;
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if(c >= B.n_chains) return;
    const int bk = B.chain_bucket[c];
    if(bk < 0) { if(B.chain_row) B.chain_row[c] = -1; return; }                      // filtered out: takes no further part, holds no column row
    const int pos = atomicAdd(&B.order_hist[bk], 1);
    B.chain_order[pos] = c;
    if(B.chain_row) B.chain_row[c] = pos;   // column rows are in position order (batch.h: row_base)
}

// 4-bit packed bases of a window (hlala_batch_in::read_bases_packed: BAM nibble codes, every read on a byte boundary at (base offset + read number + 1) >> 1) ->
// the ASCII array the kernels read.  One thread per base pair: two characters from one byte.
__global__ void k_unpack_bases(const uint8_t* __restrict__ packed, const int* __restrict__ read_off, const int n_reads, const long long rb0, const long long firstRead,
                               const long long p0, uint8_t* __restrict__ out)
{
    const int r = blockIdx.x;
    if(r >= n_reads) return;
    const int o0 = read_off[r], len = read_off[r + 1] - o0;
    const long long at = ((rb0 + (long long)o0 + firstRead + (long long)r + 1) >> 1) - p0;
    // "=ACMGRSVTWYHKDBN" in four words, first character in the low byte
    auto dec = [](unsigned c) -> uint8_t {
        const u32 w = c < 4 ? 0x4D43413Du : (c < 8 ? 0x56535247u : (c < 12 ? 0x48595754u : 0x4E42444Bu));
        return (uint8_t)(w >> (8 * (c & 3)));
    };
    for(int j = 2 * (int)threadIdx.x; j < len; j += 2 * (int)blockDim.x) {
        const unsigned b = packed[at + (j >> 1)];
        out[o0 + j] = dec(b >> 4);
        if(j + 1 < len) out[o0 + j + 1] = dec(b & 15u);
    }
}

}  // namespace hlala
