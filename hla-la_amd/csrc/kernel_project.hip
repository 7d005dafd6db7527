// kernel_project.hip -- stage A (placeholder until the projection kernel lands)
#include "batch.h"
namespace hlala {
__host__ __device__ inline size_t proj_slab_bytes(int stride, int maxNodesPerLevel) { return 256; }
__global__ void k_filter_chains(DevGraph G, DevBatch B, const long long* contig_off, const int* contig_level) {}
__global__ void k_project_chains(DevGraph G, DevBatch B, const long long* contig_off, const uint8_t* contig_seq, const int* contig_level,
                                 char* slabs, size_t slabBytes) {}
}
