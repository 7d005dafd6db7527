// batch.h -- device-resident batch of read pairs / chains (SoA, fixed column stride per chain).
#pragma once
#include "device_common.h"

namespace hlala {

struct DevBatch {
    int n_pairs, n_reads, n_chains, stride;
    int from_seeds;                 // 1: seed chains were uploaded directly (no stage A / C inputs)
    int unpaired;                   // 1: one read per unit (long-read / unpaired mode): n_reads == n_pairs, no extension DP
    // ---- inputs
    const int* read_off;            // [n_reads+1]
    const uint8_t* read_bases;      // alignment orientation of the primary
    const uint8_t* read_quals;
    const int* chain_off;           // [n_reads+1]
    const int* read_primary;        // [n_reads]
    const int* chain_read;          // [n_chains]
    const int* chain_contig;
    const int* chain_pos;
    const int* chain_offset;
    const int* chain_as;
    const uint8_t* chain_reverse;
    const int* cigar_off;
    const u32* cigar;
    // ---- stage A: seed chains (verboseSeedChain after alignment2Chain)
    int* seed_status;               // [n_chains] HLALA_CHAIN_*
    int* seed_ncols;
    int* seed_begin;
    int* seed_end;
    int* seed_removed;
    int* seed_level;                // [n_rows*stride]: the column rows of a chain start at row_base(B, chain)
    int* seed_edge;
    uint8_t* seed_g;
    uint8_t* seed_s;
    // ---- stage B: extended chains
    int* ext_status;
    int* ext_ncols;
    int* ext_begin;
    int* ext_end;
    double* ext_ll;
    int* dp_iters;                  // [2*n_chains]
    int* dp_score;                  // [2*n_chains]
    int* dp_ncols;                  // [2*n_chains] extension columns of the DP (-1: no extension)
    int* dp_sb;                     // [2*n_chains] read interval covered by the extension
    int* dp_se;
    int* dp_err;                    // [2*n_chains] 0, kernel line of a capacity failure, or -1000000 - columns
    int* dp_alias_head;             // [2*n_chains] items whose DP starts from the same cell of the same read as this item's: head of the list (-1: none)
    int* dp_alias_next;             // [2*n_chains] ... next entry of the list an item is on
    int* ext_level;                 // [n_rows*stride]
    int* ext_edge;
    uint8_t* ext_g;
    uint8_t* ext_s;
    uint8_t* ext_fromseed;
    int* ext_firstlast;             // [n_chains*4]: first level, second level, last level, second-last level (-1 = none)
    // ---- stage C: selected pair
    int* pair_status;               // [n_pairs]
    int* best_chain;                // [n_reads]
    int* n_comb;                    // [n_pairs]
    double* pair_ll;
    double* pair_mapq;
    double* mate_mapq;              // [n_reads]
    uint8_t* strands_valid;
    uint8_t* sel_mapq;              // [n_reads*stride] mapQ_perPosition of the selected chain
    // ---- counters (device): see hlala_batch_stats
    u64* counters;                  // [32]
    int* work_counter;              // [WC_N] dynamic work distribution: [0] stage A, [2] stage C, [7] chains stitched, [8]/[9] left / right DP items,
                                    //      [1]/[10] left / right items fetched, [12..35] retry lists (count, fetched) per tier 1..6 and direction
    int* retry_list;                // [16*n_chains] DP items that outgrew a capacity class: (tier 1..6) x (left, right) x n_chains; rows 14 / 15: the fail-over lists of the band kernels and the
                                    //      jump-free instantiation (WC_FO_COUNT)
    int* pair_multi;                // [2 passes][2 classes][n_pairs] pairs with several combinations, listed by k_pair_chains for k_pair_multi (kernel_pair.hip; counts in work_counter[WC_PAIR_MULTI ...])
    uint8_t* pair_deferred;         // [n_pairs] 1: a DP call of the pair went to the in-memory class; with the fused entry point its chains are stitched and
                                    //           the pair is scored in a second pass, after that class (which runs on a side stream next to the first pass)
    // ---- position order of the chains (kernel_order.hip): the kernels that walk the graph take their chains in this order, so that the work in flight at
    //      one time sits on neighbouring levels and its share of the graph arrays stays in the L2 (input order = read-name order = random positions)
    int* chain_order;               // [n_chains] chain numbers sorted by position bucket; null: input order
    int* chain_bucket;              // [n_chains] position bucket = first level >> order_shift (the last bucket: chains filtered out)
    int* order_hist;                // [order_nb + 1] bucket counts -> bucket starts -> scatter cursors
    int order_shift, order_nb;
    int long_chunk_nodes, long_max_segs;      // long-read layout, level-by-level form of the re-threading DP (kernel_project.hip): nodes a chunk of levels may hold (<= RT_SN) and long segments a read may
                                    // have beside its short ones (<= PROJL_LONGSEG); HLALA_LONG_CHUNK_NODES / HLALA_LONG_MAXSEGS shrink them so that small tests walk every branch
    int order_cost;                 // long-read layout: buckets by DESCENDING size of the chain's window (nodes between its first and last level) instead of position -- the reads that
                                    // cross a gene window take a hundred times a backbone read's time and must not be the last ones a wavefront draws (kernel_project.hip: k_filter_chains)
    // ---- column rows only for the chains that passed the filters (round 5): two thirds of a batch's chains end at k_filter_chains (strand, duplicate coordinates,
    //      processBAM.cpp:3200-3240) and never hold a column.  The filter and the position order run when the batch is CREATED (they read inputs only), the count of the
    //      ordered chains comes back with the upload's synchronisation, and the column arrays (seed_* / ext_*: 20 bytes per column slot) are sized by it: row k belongs to
    //      chain_order[k] -- rows are in position order -- and chain_row is the inverse (-1: no row).  null: row = chain number (batches made from seeds, no position order).
    int* chain_row;                 // [n_chains]
    int n_rows;
    int* dp_blk;                    // [DPL_N * dp_nblk + 1] items per block of k_dp_items and list (band left / right, jump-free left / right, general left / right); after the scan: where they start
    int* dp_list;                   // [2*n_chains] the ten dense lists of the first DP classes: slots of dp_items in position order (k_dp_lists)
    int dp_nblk;                    // blocks of k_dp_items
    int dp_band;                    // > 0: calls whose reach (read bases left + dp_band - 1 levels) stays inside a linear run of the graph go to the band kernel's lists (kernel_dp_band.hip); 0: HLALA_DP_BAND=0
    int dp_band2;                   // > 0: calls whose track run (FlatGraph::trk_out / trk_in: levels of one or two nodes) covers read bases left + dp_band2 - 1 levels go to the two-track band kernels (kernel_dp_band2.hip); 0: HLALA_DP_BAND2=0
    int dp_band2_maxj;              // ... and whose read bases left do not exceed this (HLALA_DP_BAND2_MAXJ: 15 / 31 / 63 = the 16- / 32- / 64-lane instantiation and below)
    int dp_band_risky;              // tests (HLALA_DP_BAND_RISKY=1): a call is listed for the band kernel as soon as the linear run covers its read bases -- many then walk past it and exercise the fail-over
    int dp_jf;                      // > 0: calls that meet no gap-path jump go to the lists of the jump-free instantiations, reach = read bases left + dp_jf - 1 levels (0: HLALA_DP_JF=0, every call in the general one)
    void* dp_items;                 // [2*n_chains] DpItem (kernel_dp.hip)
    int* dbg;                       // non-null with HLALA_DEBUG=1: kernels add phase clocks to counters[16..31]
};

// number of entries of B.chain_order: the chains that passed the filters (k_filter_chains); without a position order every chain is listed.
// (order_hist[order_nb - 1] is the start of a bucket nothing is put into = the end of the last real bucket, before and after the scatter)
__device__ __forceinline__ int ordered_chains(const DevBatch& B) { return B.chain_order ? __builtin_amdgcn_readfirstlane(B.order_hist[B.order_nb - 1]) : B.n_chains; }

// first column slot of a chain's rows in seed_* / ext_* (only chains that passed the filters have rows: callers look at the chain's status first)
__device__ __forceinline__ size_t row_base(const DevBatch& B, int c) { return (size_t)(B.chain_row ? B.chain_row[c] : c) * (size_t)B.stride; }

// ---- the first DP classes' dense item lists (k_dp_items / k_dp_lists): list k occupies dp_list[dp_blk[k * dp_nblk] .. dp_blk[(k + 1) * dp_nblk]); k + 1: the right extensions
enum { DPL_BAND16 = 0, DPL_BAND32 = 2, DPL_BAND64 = 4, DPL_JF = 6, DPL_GEN = 8, DPL_B2_16 = 10, DPL_B2_32 = 12, DPL_B2_64 = 14, DPL_N = 16 };
// the list an item is put on, by the class k_dp_items gives it (0 general, 1 jump-free, 2 / 3 / 4 band kernel with 16 / 32 / 64 lanes per call, 5 / 6 / 7 two-track band kernel); + 1 for a right extension
__host__ __device__ inline int dpl_of_class(int cls) { return cls == 0 ? DPL_GEN : (cls == 1 ? DPL_JF : (cls == 2 ? DPL_BAND16 : (cls == 3 ? DPL_BAND32 : (cls == 4 ? DPL_BAND64 : (cls == 5 ? DPL_B2_16 : (cls == 6 ? DPL_B2_32 : DPL_B2_64)))))); }
// ---- B.work_counter (WC_N ints): [0] stage A, [1] / [10] left / right general items fetched, [2] stage C, [4] / [5] jump-free items fetched, [6] jump-free calls (statistics),
// [7] chains stitched, [8] / [9] left / right DP calls, [12..35] retry lists of tiers 1..6 (count, fetched) x (left, right), [36] / [37] second stitch / pairing pass,
// [40..47] round 6: the lists of the pairs with several combinations (k_pair_chains -> k_pair_multi): per pass (main / side stream) count and fetched of class 0, of class 1; round 5: items fetched by the three band kernels (left, right each), the fail-over list's counts and fetch counters, band calls that
// failed over, band calls listed, jump-free calls that met a jump, and why band calls failed ([WC_BAND_WHY + 2 .. + 5]: past the staged levels, past the linear run, too
// many iterations, too many tied end cells)
// round 6: [72..77] items fetched by the three two-track band kernels (left, right each), [78] calls listed for them, [79] those that failed over, [80..87] why (2 end of the
// staged track steps, 4 too many iterations, 5 too many tied end cells, 6-8 internal)
enum { WC_PAIR_MULTI = 40, WC_BAND_FETCH = 48, WC_FO_COUNT = 54, WC_FO_FETCH = 56, WC_BAND_FAILED = 58, WC_BAND_CALLS = 59, WC_JF_FAILED = 60, WC_BAND_WHY = 62, WC_BAND_TIED = 68,
       WC_B2_FETCH = 72, WC_B2_CALLS = 78, WC_B2_FAILED = 79, WC_B2_WHY = 80, WC_N = 88 };
// the fail-over list of the first classes: calls the band kernel (kernel_dp_band.hip) or the jump-free instantiation could not finish; k_dp<DpTiny, 0> draws it after
// its own lists.  Entries (slots of dp_items) at retry_list[(14 + direction) * n_chains ...], counts in work_counter[WC_FO_COUNT + direction].
// ---- capacities of the band kernels (kernel_dp_band.hip): read bases a call may have left for the instantiation with 16 / 32 / 64 lanes per call; k_dp_items lists a call
// for one of them when the linear run ahead of its start level covers bases left + margin + min(bases left + 6, 40) levels
constexpr int BAND_MAXJ16 = 15, BAND_MAXJ32 = 31, BAND_MAXJ64 = 48;
// ... of the two-track band kernels (kernel_dp_band2.hip): read bases per instantiation, iterations a call may run (rows of a wavefront's back-pointer slab)
constexpr int B2_MAXJ16 = 15, B2_MAXJ32 = 31, B2_MAXJ64 = 63, B2_MAXD = 256;

enum {
    CNT_CHAINS_EXT = 0, CNT_DP_CALLS, CNT_DP_ITERS, CNT_DP_CELLS, CNT_SEED_COLS, CNT_OUT_COLS, CNT_EDGES, CNT_ERRORS, CNT_DP_SHARED
};

}  // namespace hlala
