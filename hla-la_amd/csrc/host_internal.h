// host_internal.h -- shared between the host-side translation units of the library (not part of the ABI)
#ifndef HLALA_HOST_INTERNAL_H_
#define HLALA_HOST_INTERNAL_H_
#include <string>
#include <utility>
#include <vector>

#include "../../include/hlala_gpu.h"

namespace hlala_host {

// one allele of one exon position as counted before the high-coverage / strand filters (hla/HLATyper.cpp:1731-1786)
struct AlleleTally {
    std::string allele;
    int count = 0, reverse = 0, from_first = 0;
    int post_filtering = -1;       // perPosition_allele_counts_postFiltering (:1821); -1 = no entry
};

// hlala_filter_positions; tallies (optional) is indexed by exon position, alleles in order of first appearance
int filter_positions_impl(const hlala_exon_positions_out* pos, const hlala_filter_params* prm, uint8_t* pos_use, uint8_t* read_ignored, hlala_filter_stats* stats,
                          std::vector<std::vector<AlleleTally>>* tallies);

// the bulk arrays of a seed batch (host_bam.cpp) for hlala_seed_batch_pin, its pinned flag, and the hook hlala_seed_batch_free calls for a pinned batch
void seed_batch_bulk_arrays(hlala_seed_batch* S, std::vector<std::pair<void*, size_t>>& out);
bool& seed_batch_pinned_flag(hlala_seed_batch* S);
extern void (*g_seed_batch_unpin)(hlala_seed_batch*);

}  // namespace hlala_host
#endif
