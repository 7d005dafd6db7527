// host_internal.h -- shared between the host-side translation units of the library (not part of the ABI)
#ifndef HLALA_HOST_INTERNAL_H_
#define HLALA_HOST_INTERNAL_H_
#include <string>
#include <utility>
#include <vector>
#include <mutex>

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

// CPUs this process may keep busy: the hardware threads, or fewer under a CFS quota of its control group (cgroup v2 cpu.max, v1 cpu.cfs_quota_us) -- a
// container on a 256-thread host may own 16 of them, and threads beyond twice the quota only get each other throttled (measured: the decoder of an
// 8.4 M-pair sample on such a host takes 5.2 / 3.5 / 2.85 / 3.0 / 3.3-3.7 s on 8 / 16 / 32 / 64 / 128 threads)
int host_cpu_budget();

// the bulk arrays of a seed batch (host_bam.cpp) for hlala_seed_batch_pin, its pinned flag, and the hook hlala_seed_batch_free calls for a pinned batch
void seed_batch_bulk_arrays(hlala_seed_batch* S, std::vector<std::pair<void*, size_t>>& out);
void seed_batch_bulk_arrays(hlala_seed_batch* S, std::vector<std::pair<void*, size_t>>& out, int64_t unit_end, std::vector<size_t>* upto);       // ... + bytes of each that the units before unit_end occupy
bool& seed_batch_pinned_flag(hlala_seed_batch* S);
bool& seed_batch_pin_lazy(hlala_seed_batch* S);
std::mutex& seed_batch_pin_mutex(hlala_seed_batch* S);
std::vector<size_t>& seed_batch_pin_cursor(hlala_seed_batch* S);
std::vector<std::pair<void*, size_t>>& seed_batch_pin_regions(hlala_seed_batch* S);
extern void (*g_seed_batch_pin_upto)(hlala_seed_batch*, int64_t);
extern void (*g_seed_batch_unpin)(hlala_seed_batch*);

}  // namespace hlala_host
#endif
