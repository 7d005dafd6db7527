// kernel_pair.hip -- stage C (placeholder until the pairing kernel lands)
#include "batch.h"
namespace hlala {
__global__ void k_pair_chains(DevGraph G, const DevTables* Tp, DevBatch B) {}
}
