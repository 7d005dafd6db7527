/*
 * hlala_gpu.h -- C ABI of libhlala_gpu.so, the MI355X (gfx950) implementation of HLA*LA's
 * read-to-PRG alignment hot path.
 *
 * The reference (DiltheyLab/HLA-LA) has no FFI seam: the path is reached through C++ member
 * calls inside one binary.  Every entry point below therefore names the reference member
 * function(s) it stands in for (file:line in the reference tree), so that a maintainer can
 * swap the call (see INTEGRATION.md for the binding code).
 *
 * Conventions
 *  - plain C, no exceptions cross the boundary; every function returns 0 on success and a
 *    negative HLALA_E_* code on failure; hlala_last_error() gives the text.
 *  - the caller owns every host buffer; the library owns device memory behind the handles.
 *  - one hlala_ctx per GPU/process; calls on one ctx are serialised by the caller.
 *  - all indices are 0-based; "level" is a PRG graph level as in the reference
 *    (Graph::NodesPerLevel index); level -1 marks a read base aligned against nothing.
 *  - node rank z within a level, edge order within a node and jump order are CANONICAL:
 *    creation order in graph.txt (the reference uses heap-pointer order).
 */
#ifndef HLALA_GPU_H_
#define HLALA_GPU_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HLALA_OK              0
#define HLALA_E_ARG          -1   /* bad argument / inconsistent sizes              */
#define HLALA_E_DEVICE       -2   /* HIP runtime error, no device, no kernel image  */
#define HLALA_E_GRAPH        -3   /* graph violates the levelled-DAG invariants     */
#define HLALA_E_CAPACITY     -4   /* a fixed device capacity was exceeded           */
#define HLALA_E_STATE        -5   /* call sequence error (stage not run yet, ...)   */

/* per-chain status (hlala_chains_out.status) */
#define HLALA_CHAIN_OK             0  /* projected, extended and scored                             */
#define HLALA_CHAIN_SKIP_STRAND    1  /* strand differs from the primary's (processBAM.cpp:3216)     */
#define HLALA_CHAIN_SKIP_DUP       2  /* same start//stop id seen with >= AS (processBAM.cpp:3234)   */
#define HLALA_CHAIN_ERR_COLUMNS   -1  /* more alignment columns than params.max_columns              */
#define HLALA_CHAIN_ERR_FRONTIER  -2  /* DP frontier / candidate / cell capacity exceeded            */
#define HLALA_CHAIN_ERR_INPUT     -3  /* CIGAR/coordinates the reference would assert or throw on    */

/* ------------------------------------------------------------------------------------------
 * Graph: replaces Graph::readFromFile's in-memory result (Graph/Graph.cpp:2329-2559) as input to
 * alignerBase::alignerBase (mapper/aligner/alignerBase.cpp:16-38), Graph::computeGapEdgePaths
 * (Graph/Graph.cpp:347-476) and the gap-stretch scan of processBAM::processBAM
 * (mapper/processBAM.cpp:91-149).  Arrays are in graph.txt creation order.
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    int32_t        n_levels;    /* number of node levels L; edges connect level l -> l+1      */
    int32_t        n_nodes;
    int32_t        n_edges;
    const int32_t* node_level;  /* [n_nodes]                                                  */
    const int32_t* edge_from;   /* [n_edges] node index                                       */
    const int32_t* edge_to;     /* [n_edges] node index                                       */
    const uint8_t* edge_label;  /* [n_edges] decoded emission char: 'A','C','G','T','N','_','*' */
} hlala_graph_desc;

/* Linear reference contigs the seeds were aligned to, with their position->level tables:
 * replaces processBAM::{extendedReferenceGenomeSequences,PRGonlyReferenceGenomeSequences} and
 * processBAM::_loadMapping (mapper/processBAM.cpp:4389-4457).  All tables are loaded up front
 * (SURVEY.md Appendix A3 note on load order). */
typedef struct {
    int32_t        n_contigs;
    const int64_t* contig_off;   /* [n_contigs+1] offsets into contig_seq / contig_level       */
    const uint8_t* contig_seq;   /* bases                                                      */
    const int32_t* contig_level; /* translation/<SequenceID>.txt: graph level of each position */
    const int32_t* contig_seqid; /* [n_contigs] PRG SequenceID (key order of the insert-size scan) */
} hlala_contigs_desc;

typedef struct {
    double   insert_mean;        /* boost::math::normal(IS_mean, IS_sd), processBAM.cpp:2342    */
    double   insert_sd;
    uint32_t rng_seed;           /* base of extensionAligner::rng_seeds (extensionAligner.h:38):
                                    the DP of chain c, direction d (0 = left, 1 = right) starts
                                    from seed rng_seed + 2*c + d, c = absolute chain index       */
    int32_t  long_read_mode;     /* 0: indel rate 0.001; !=0: 0.075 (extensionAligner.cpp:58-64) */
    int32_t  max_columns;        /* per-chain alignment column capacity (stride of col_* arrays) */
    int32_t  reserved;
} hlala_params;

typedef struct hlala_ctx   hlala_ctx;
typedef struct hlala_batch hlala_batch;

/* Build the device-resident CSR (levels, per-node ordered edge lists, gap-path jump tables,
 * gap-stretch bitmap, level->(sequence,position) table) once.  `stream` is a hipStream_t (may
 * be NULL for the default stream); all kernels of this ctx are launched on it.              */
int  hlala_create(hlala_ctx** out, int device, void* stream,
                  const hlala_graph_desc* graph, const hlala_contigs_desc* contigs,
                  const hlala_params* params);
void hlala_destroy(hlala_ctx* ctx);
const char* hlala_last_error(const hlala_ctx* ctx);   /* ctx may be NULL: create-time errors */

/* Flattened-graph introspection (parity of the one-time host pass with the oracle).          */
typedef struct {
    int32_t n_levels, n_nodes, n_edges, n_paths;
    int64_t n_jump_entries, n_path_edges, n_levelpos_entries;
    int32_t max_nodes_per_level, max_out_degree, max_in_degree, n_gap_stretch_levels;
    int32_t max_jumps, max_parallel;   /* most gap-path jumps of a node in one direction; most parallel edges / gap paths between one pair of nodes */
} hlala_graph_info;
int hlala_graph_get_info(const hlala_ctx* ctx, hlala_graph_info* info);
/* node renumbering: device node id -> creation index, [n_nodes]; level offsets [n_levels+1] */
int hlala_graph_get_nodes(const hlala_ctx* ctx, int32_t* node_orig, int32_t* level_off);
/* completed gap-edge paths in completedGapEdgePaths order (Graph.cpp:417): first/last node
 * (creation idx) and length, [n_paths] each; path_edges may be NULL                         */
int hlala_graph_get_paths(const hlala_ctx* ctx, int32_t* first_node, int32_t* last_node,
                          int32_t* length);
int hlala_graph_get_gap_stretch(const hlala_ctx* ctx, uint8_t* in_stretch /* [n_levels-1] */);

/* ------------------------------------------------------------------------------------------
 * One batch of read pairs with their BWA alignments ("proto seeds", mapper/reads/protoSeeds.h):
 * read 2p is mate 1 of pair p, read 2p+1 is mate 2.  Bases/qualities are those of the mate's
 * PRIMARY alignment in alignment orientation (processBAM.cpp:3142-3145); chains of a read are
 * its BAM records in AS-descending order (processBAM.cpp:1945).
 *
 * A batch is a WINDOW into the arrays of a sample: the three offset arrays are 64-bit and need not start at 0.
 * Read r of the batch has its bases at read_bases[read_off[r] .. read_off[r+1]) and its chains at index
 * [chain_off[r], chain_off[r+1]) of every chain_* array (read_primary holds indices of the same numbering, cigar_off is
 * indexed by it).  A caller that holds a whole sample (BASELINE config 3: ~10 M pairs = 3e9 bases, beyond 32-bit offsets)
 * passes `read_off + 2*u0`, `chain_off + 2*u0`, `read_primary + 2*u0` and n_pairs = n for the units [u0, u0 + n) and leaves
 * every other pointer alone (hlala_seed_batch_window does exactly that); the library uploads the window and rebases it.
 * chain_off[0] is taken as the batch's first absolute chain number (hlala_batch_set_first_chain).  ONE batch is limited to
 * 2^31-1 bases / chains / CIGAR operations (HLALA_E_CAPACITY beyond that: cut the sample into more batches).
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    int32_t         n_pairs;
    const int64_t*  read_off;      /* [2n+1] offsets into read_bases / read_quals               */
    const uint8_t*  read_bases;    /* ASCII                                                     */
    const uint8_t*  read_quals;    /* ASCII Phred+33                                            */
    const int64_t*  chain_off;     /* [2n+1] chains of read r: [chain_off[r], chain_off[r+1])   */
    const int32_t*  read_primary;  /* [2n] chain index (numbering of chain_off) of the read's primary alignment */
    int32_t         n_chains;      /* chain_off[2n] - chain_off[0]                              */
    const int32_t*  chain_contig;  /* [chain index] index into hlala_contigs_desc               */
    const int32_t*  chain_pos;     /* BamAlignment::Position (0-based leftmost)                 */
    const int32_t*  chain_offset;  /* reference2level_offset (interval start)                   */
    const int32_t*  chain_as;      /* AS tag                                                    */
    const uint8_t*  chain_reverse; /* IsReverseStrand()                                         */
    const int64_t*  cigar_off;     /* [chain index .. +1] offsets into cigar                    */
    const uint32_t* cigar;         /* BAM encoding: len<<4 | op, op index into "MIDNSHP=X"      */
    /* Optional (round 4): the bases 4-bit packed as a BAM record holds them (codes of "=ACMGRSVTWYHKDBN", two per byte, the first base in the high nibble) --
     * half the bytes to upload, and what a BAM decoder has without unpacking.  When read_bases_packed is not NULL, read_bases is not read.  Every read starts on a
     * byte: read R of the SAMPLE (R = first_read + r for read r of the window) has its (length + 1) / 2 bytes at read_bases_packed[(read_off[r] + R + 1) >> 1 ...],
     * a closed form in the 64-bit base offset and the read's number that leaves room for one spare nibble per read (hlala_pack_bases writes this layout,
     * hlala_seed_batch_window of a sample decoded with HLALA_SEEDS_PACKED hands it out).  The library unpacks on the device.                                       */
    const uint8_t*  read_bases_packed;
    int64_t         first_read;    /* number of the window's first read in the sample (0 for a batch that is its own sample) */
} hlala_batch_in;
/* ASCII bases of reads [0, n_reads) with offsets read_off (starting at 0) -> the packed layout above; `packed` holds (read_off[n_reads] + n_reads + 1) / 2 + 1 bytes.
 * Characters outside "=ACMGRSVTWYHKDBN" are packed as N. */
int  hlala_pack_bases(const uint8_t* read_bases, const int64_t* read_off, int64_t n_reads, uint8_t* packed);

/* Seed chains handed over directly, bypassing the BAM projection: the protocol of
 * `--action testChainExtension` (HLA-LA.cpp:1733-1861) and the input type of
 * extensionAligner::extendSeedChain (mapper/aligner/extensionAligner.cpp:186).  Chain c
 * belongs to read chain_read[c]; columns of chain c are [col_off[c], col_off[c+1]).          */
typedef struct {
    int32_t         n_reads;
    const int32_t*  read_off;       /* [n_reads+1] */
    const uint8_t*  read_bases;
    const uint8_t*  read_quals;
    int32_t         n_chains;
    const int32_t*  chain_read;     /* [n_chains] */
    const int32_t*  chain_seq_begin;/* verboseSeedChain::sequence_begin */
    const int32_t*  chain_seq_end;  /* verboseSeedChain::sequence_end   */
    const uint8_t*  chain_reverse;
    const int32_t*  col_off;        /* [n_chains+1] */
    const int32_t*  col_level;      /* graph_aligned_levels */
    const int32_t*  col_edge;       /* graph_aligned_edges as edge creation index, -1 = none */
    const uint8_t*  col_gchar;      /* graph_aligned    */
    const uint8_t*  col_schar;      /* sequence_aligned */
} hlala_seeds_in;

/* Chain-level results (mapper::reads::verboseSeedChain, verboseSeedChain.h:22-50), fixed
 * stride = params.max_columns columns per chain.  Any pointer may be NULL to skip it.       */
typedef struct {
    int32_t* status;        /* [n_chains] HLALA_CHAIN_*                                      */
    int32_t* n_cols;        /* [n_chains]                                                    */
    int32_t* seq_begin;     /* [n_chains]                                                    */
    int32_t* seq_end;       /* [n_chains]                                                    */
    int32_t* removed_cols;  /* [n_chains] removed_columns_noGap_restriction (seed stage)     */
    double*  ll;            /* [n_chains] scoreOneAlignment (extended stage only)            */
    int32_t* dp_iters;      /* [2*n_chains] DP iterations run, left/right (extended stage)   */
    int32_t* dp_score;      /* [2*n_chains] score of the DP end cell, INT32_MIN = no extension */
    int32_t* col_level;     /* [n_chains*stride]                                             */
    int32_t* col_edge;      /* [n_chains*stride] edge creation index, -1 = none              */
    uint8_t* col_gchar;     /* [n_chains*stride]                                             */
    uint8_t* col_schar;     /* [n_chains*stride]                                             */
    uint8_t* col_fromseed;  /* [n_chains*stride] is_from_BWAseed                             */
} hlala_chains_out;

/* Pair-level results (mapper::reads::verboseSeedChainPair, verboseSeedChain.h:318-346):
 * the selected chain of each mate with mapping qualities.  Any pointer may be NULL (that array is then not
 * transferred); of the column arrays only the first n_cols entries of a row are meaningful, the rest come back zero. */
typedef struct {
    int32_t* pair_status;   /* [n] 0 ok, <0: a chain of the pair hit an HLALA_CHAIN_ERR_*    */
    int32_t* best_chain;    /* [2n] absolute chain index selected for each mate              */
    int32_t* n_combinations;/* [n] read1_extendedChains.size()*read2_extendedChains.size()   */
    double*  pair_ll;       /* [n] combinations_max.first (processBAM.cpp:3538)              */
    double*  pair_mapq;     /* [n] verboseSeedChainPair::mapQ                                */
    double*  mate_mapq;     /* [2n] verboseSeedChain::mapQ                                   */
    uint8_t* strands_valid; /* [n] alignedReadPair_strandsValid of the selected pair         */
    int32_t* n_cols;        /* [2n]                                                          */
    int32_t* col_level;     /* [2n*stride]                                                   */
    int32_t* col_edge;      /* [2n*stride]                                                   */
    uint8_t* col_gchar;     /* [2n*stride]                                                   */
    uint8_t* col_schar;     /* [2n*stride]                                                   */
    uint8_t* col_fromseed;  /* [2n*stride]                                                   */
    uint8_t* col_mapq;      /* [2n*stride] mapQ_perPosition (Phred char)                     */
} hlala_pairs_out;

/* Upload a batch: inputs become resident in HBM; nothing is computed.  Paired reads of more than 1024 bases are refused
 * (HLALA_E_CAPACITY): the extension DP keys its cells with a 12-bit read coordinate; long reads go through hlala_batch_create_unpaired.
 * Returns when the caller's buffers have been read.  The OUTPUT arrays of the batch (50 GB per million pairs) are allocated by its first stage call
 * (hlala_project_chains / hlala_align_batch), not here: a batch that is only uploaded costs its inputs.
 * The checks of the offsets (the reference's asserts) and their rebasing run on up to four host threads into page-locked scratch of the context: 5 ms for a batch of a million pairs
 * (round 6; 30 ms on the calling thread before).  One caller thread per context at a time. */
int  hlala_batch_create(hlala_ctx* ctx, const hlala_batch_in* in, hlala_batch** out);
/* Long-read / unpaired mode (processBAM::alignOneLongRead, mapper/processBAM.cpp:3618-3838, and
 * assignMappingQualities_unpaired, :3900-4059): `in->n_pairs` is the number of READS, every array that is per read
 * has one entry per unit (read_off[n+1], chain_off[n+1], read_primary[n]).  hlala_align_batch then projects every
 * alignment that passes the strand / duplicate filters, pads it to the full read (no extension DP: :3733-3735),
 * scores it (long-read rates if params.long_read_mode), selects the first maximum and assigns the unpaired mapping
 * qualities.  hlala_pairs_out arrays are per unit: [n] where the paired layout has [2n].  params.max_columns bounds
 * the columns of a chain (up to 16384: kilobase reads; above 512 the projection works out of HBM scratch instead of
 * LDS); longer chains (at any stage of the projection: the columns padded in for skipped graph levels count, so leave headroom) are flagged HLALA_CHAIN_ERR_COLUMNS, as are reads with several alignments AND more than 512
 * columns (the reference keeps primaries only in long-read mode, processBAM.cpp:725-738).                            */
int  hlala_batch_create_unpaired(hlala_ctx* ctx, const hlala_batch_in* in, hlala_batch** out);
/* Upload seed chains directly (stage A is then not available on this batch).                */
int  hlala_batch_create_from_seeds(hlala_ctx* ctx, const hlala_seeds_in* in, hlala_batch** out);
/* A sample that is pushed through the GPU in several batches (BASELINE config 3: ~10 M pairs) keeps ONE numbering of its chains:
 * `first_chain` is the absolute index of this batch's chain 0, so that the DP of chain c, direction d draws from
 * rng_seed + 2*(first_chain + c) + d -- the seeds of the unsplit run (params.rng_seed above; the reference's rng_seeds is a public
 * mutable member the caller sets per call, extensionAligner.h:38).  Default 0.  Call before hlala_extend_chains / hlala_align_batch. */
int  hlala_batch_set_first_chain(hlala_batch* b, uint32_t first_chain);
/* The device buffers are parked in the context and reused by the next batch (a batch that outlives its context frees them itself)
 * of similar size (allocating the column arrays of a 1 M-pair batch costs about a second otherwise).                 */
void hlala_batch_destroy(hlala_batch* b);

/* Stage A -- processBAM::alignment2Chain (mapper/processBAM.cpp:3019-3127) for every chain that
 * survives the strand / duplicate-coordinate filters of alignOneReadPair (:3200-3240):
 * transformBAMreadToInternalAlignment (:4794), PRGContigAlignment2Seed (:2491) incl.
 * cleanInitialAlignment (:4621) and restrictInitialAlignmentToNoGapAreas (:4461).           */
int  hlala_project_chains(hlala_ctx* ctx, hlala_batch* b);
/* Stage B -- extensionAligner::extendSeedChain (mapper/aligner/extensionAligner.cpp:186-333)
 * = fullNeedleman_diagonal_extension_gapJumper (:335-1556) left and right, stitching
 * (verboseSeedChain.cpp:23-136), then extensionAligner::scoreOneAlignment (:52-182).         */
int  hlala_extend_chains(hlala_ctx* ctx, hlala_batch* b);
/* Stage C -- the pairing loop of processBAM::alignOneReadPair (mapper/processBAM.cpp:3408-3546)
 * and processBAM::assignMappingQualities (:4062-4312).                                       */
int  hlala_pair_chains(hlala_ctx* ctx, hlala_batch* b);
/* A + B + C: processBAM::alignOneReadPair (mapper/processBAM.cpp:3129-3616) over the batch.
 * Asynchronous like the stage calls.  On a paired batch the few percent of DP calls with wide frontiers (allele-rich gene levels) and the
 * pairs that own them finish on a second stream of the context, beside the rest of the batch; every later call that touches THIS batch
 * (getters, statistics, post-processing, the next alignment of it, its destruction) is ordered behind that work.  A caller with several
 * batches can therefore keep two in flight -- align A, align B, fetch A, align C, fetch B, ... -- and have the long tail of one batch
 * run beside the bulk of the next; same results as one batch at a time.                                                     */
int  hlala_align_batch(hlala_ctx* ctx, hlala_batch* b);
/* The tail pool (round 6).  0.07 % of a batch's DP calls -- frontiers of hundreds to thousands of cells on allele-rich levels -- run in three capacity classes
 * whose cost is the latency of their slowest calls, paid once per LAUNCH.  With k > 1, hlala_align_batch on a paired batch leaves those classes and the pairs that
 * wait for them PENDING; when k batches are pending (or on hlala_flush, or as soon as a call reads, re-runs or destroys a pending batch) each class is launched once
 * over all pending batches, then every batch's deferred pairs are completed.  A caller that walks a sample (mapper/processBAM.cpp:2024-2039: the loop over
 * the read pairs of a sample) keeps k + 1 batches in flight -- align i .. i + k, fetch i, align i + k + 1, ... -- and pays that latency once per k batches.
 * Results do not depend on k.  k = 1 (default): every alignment runs its own tail, as in rounds 2-5.  1 <= k <= 8.  The replaced reference code has no
 * counterpart: extensionAligner.cpp:335-1556 runs one DP call at a time on one thread.                                          */
int  hlala_set_tail_pool(hlala_ctx* ctx, int k);
int  hlala_flush(hlala_ctx* ctx);

/* Download results (synchronises the stream).  `stage` 0 = seed chains after stage A,
 * 1 = extended chains after stage B.                                                         */
int  hlala_batch_get_chains(hlala_ctx* ctx, hlala_batch* b, int stage, hlala_chains_out* out);
int  hlala_batch_get_pairs(hlala_ctx* ctx, hlala_batch* b, hlala_pairs_out* out);

/* The same columns without the padding: the selected alignments are 150-odd columns long, the rows above max_columns (384).  Columns of
 * read r (mate m of pair p: r = 2p + m; unpaired batches: r = p) sit at [col_off[r], col_off[r+1]) of every col_* array; 2.6 times
 * less data over PCIe for 2x150 bp reads.  Per-pair scalars: hlala_batch_get_pairs with NULL column pointers.  Returns HLALA_E_CAPACITY
 * with n_cols_total set when cap_cols is too small (call once with cap_cols = 0 to size the arrays). */
typedef struct {
    int64_t  cap_cols;       /* in: capacity of the col_* arrays                         */
    int64_t  n_cols_total;   /* out: columns of all selected alignments                  */
    int64_t* col_off;        /* [n_reads + 1]                                            */
    int32_t* col_level;      /* any of the col_* pointers may be NULL                    */
    int32_t* col_edge;
    uint8_t* col_gchar;
    uint8_t* col_schar;
    uint8_t* col_fromseed;
    uint8_t* col_mapq;
} hlala_pairs_packed_out;
int  hlala_batch_get_pairs_packed(hlala_ctx* ctx, hlala_batch* b, hlala_pairs_packed_out* out);

/* Device-to-device export of one fixed-size record per pair (8 doubles: pair_status, best_chain[0], best_chain[1],
 * n_combinations, pair_ll, pair_mapq, mate_mapq[0], mate_mapq[1]) into a caller-owned DEVICE buffer of
 * 8 * n_pairs doubles -- the payload of the multi-GPU gather of per-pair best-path records to rank 0
 * (the host program hands the buffer to RCCL; there is no collective inside this library).  Runs like the getters: behind
 * THIS batch's work only (not behind other batches queued since) and returns when the records are in `device_out`, so the
 * caller may start its collective on any stream at once.                                             */
int  hlala_batch_export_pair_records(hlala_ctx* ctx, hlala_batch* b, double* device_out);

/* ---- several GPUs in ONE process (the host program's --devices 0,1,...: one context per GPU).  The reference merges the results of its threads on the host
 * (mapper/processBAM.cpp:1866-1887: the per-thread vectors of aligned pairs appended, the per-level read counters added up, written out at :1902-1913); with one
 * context per GPU these are the two exchange steps of the path, over RCCL (xGMI): hlala_gather_pair_records = one grouped send / receive of the per-pair records
 * to the first context's device (the counts are known to the host, which holds every batch), hlala_reduce_coverage = ncclReduce (sum) of bases_per_level.
 * hlala_comm_create runs ncclCommInitAll over the contexts' devices when there are several, all different (librccl.so is looked up at run time); a communicator
 * of one context, or of contexts that share a device, copies device to device instead -- same results, hlala_comm_uses_rccl() says which.
 * HLALA_COMM_RCCL=1 forces RCCL for a communicator of one (tests), =0 forbids it.  The multi-PROCESS form of the same steps (one rank per GPU, torch.distributed
 * over RCCL) is hla-la_amd/dist.py. */
typedef struct hlala_comm hlala_comm;
int  hlala_comm_create(hlala_ctx* const* ctxs, int n, hlala_comm** out);
void hlala_comm_destroy(hlala_comm* comm);
int  hlala_comm_uses_rccl(const hlala_comm* comm);
const char* hlala_comm_last_error(const hlala_comm* comm);      /* NULL: the error of the last failed hlala_comm_create of this thread */
/* batches[i]: the batch of context i (NULL: none this round); host_out: [sum of pairs][8] doubles in context order (hlala_batch_export_pair_records' layout);
 * counts_out[i] (may be NULL) = pairs of context i.  Returns when host_out is filled. */
int  hlala_gather_pair_records(hlala_comm* comm, hlala_batch* const* batches, double* host_out, int64_t host_capacity_pairs, int64_t* counts_out);
/* bases_per_level[n_levels - 1] summed over the communicator's contexts (hlala_get_coverage of each, added up on the first context's device) */
int  hlala_reduce_coverage(hlala_comm* comm, int32_t* bases_per_level, int reset);

/* Per-stage statistics of the last hlala_align_batch / stage call on this batch, measured
 * with HIP events (ms) plus work counters reduced on the device.  The events belong to the BATCH (since round 3): with
 * several batches in flight every batch reports its own stages.                              */
typedef struct {
    float   ms_project, ms_extend, ms_pair;
    int64_t n_chains_extended;    /* chains with status OK                        */
    int64_t n_dp_calls;           /* DP invocations (left + right)                */
    int64_t n_dp_iterations;      /* sum of DP iterations                         */
    int64_t n_dp_cells;           /* candidate target cells evaluated             */
    int64_t n_seed_columns;       /* columns of all projected seed chains         */
    int64_t n_out_columns;        /* columns of all extended chains               */
    int64_t n_edges_touched;      /* CSR edge records read by stages A and B      */
    int64_t n_errors;             /* chains with status < 0                       */
    float   ms_extend_retry;      /* part of ms_extend spent re-running DP calls in the 64-lane classes */
    int32_t n_chains_retried;     /* re-runs of DP calls in a wider capacity class (a call can be re-run twice) */
    float   ms_dp_main;           /* part of ms_extend spent in the 16-lane DP kernel (k_dp<DpTiny>)      */
    int32_t n_dp_retried_large;   /* DP calls that also outgrew the 64-lane small class                  */
    int64_t n_dp_shared;          /* of n_dp_calls: calls that start from the same cell of the same read as the call of an earlier chain
                                     and took their iterations from it (own end-cell choice, backtrace and columns)                  */
    int32_t n_dp_class[7];        /* DP calls that entered the 16-lane / 32-lane / 64-lane / wide / broad / large / in-memory class (a call that
                                     outgrows a class enters a later one as well)                                                               */
    float   ms_dp_class[7];       /* time of the DP kernel of each class (HIP events on the stream the kernel ran on)                */
    float   ms_side;              /* hlala_align_batch on a paired batch: span of the work it put on its second stream (wide / broad / large / in-memory
                                     class, second stitch and pairing pass over the pairs that own those calls), which runs beside the rest of the
                                     batch and beside the caller's next batch; 0 when the stages were called one by one                          */
    int32_t n_dp_lane;            /* always 0 (the lane-per-DP class of round 3 lost its A/B and was removed in round 6; the two fields keep the layout)       */
    float   ms_dp_lane;           /* always 0                                                                                                                 */
    int32_t n_dp_jump_free;       /* of n_dp_class[0]: calls that were known to meet no gap-path jump and ran in the instantiation of the 16-lane class that is
                                     compiled without the early-cell machinery (kernel_dp.hip: DpTinyJF)                                                      */
    float   ms_dp_jump_free;      /* part of ms_dp_class[0] spent in that instantiation                                                                       */
    int32_t n_dp_band;            /* DP calls listed for the band kernel (kernel_dp_band.hip: the levels within reach are a linear stretch of the graph -- one node per
                                     level, one edge per step --, anti-diagonals held in registers, no cell table); they do not enter the 16-lane class ...          */
    int32_t n_dp_band_failed;     /* ... except these: calls that left the band or the staged levels and were re-run in the general 16-lane instantiation (counted in
                                     n_dp_class[0])                                                                                                                */
    int32_t n_dp_jump_free_failed;/* of n_dp_jump_free: calls that met a gap-path jump after all and were re-run in the general 16-lane instantiation                  */
    float   ms_dp_band;           /* time of the band kernel (before the 16-lane class on the main stream; not part of ms_dp_class[0])                                 */
    int32_t n_dp_band2;           /* round 6: DP calls listed for the two-track band kernels (kernel_dp_band2.hip: one or two nodes per level within reach, at most one
                                     gap-path jump -- the neighbourhood of a gap stretch); they do not enter the 16-lane class ...                                     */
    int32_t n_dp_band2_failed;    /* ... except these: calls that reached the end of their track steps and were re-run in the general 16-lane instantiation            */
    float   ms_dp_band2;          /* time of those kernels (after the band kernels, before the 16-lane class; not part of ms_dp_class[0])                              */
    int32_t reserved_stats;
} hlala_batch_stats;
int  hlala_batch_get_stats(hlala_ctx* ctx, hlala_batch* b, hlala_batch_stats* out);

/* ------------------------------------------------------------------------------------------
 * Graph files (host code, no context needed).  hlala_graph_load_text parses PRG/graph.txt in the format of
 * Graph::writeToFile / Graph::readFromFile (Graph/Graph.cpp:2225-2327 / :2329-2545; allele codes:
 * Graph/LocusCodeAllocation.cpp:264-312, deCode :32-48); nodes and edges keep their order in the file, node indices are
 * renumbered 0.. in that order.  hlala_graph_cache_save / _load keep the same arrays in a binary file: loading is a few
 * reads instead of minutes of text (or Boost archive) parsing.  hlala_graph_file_desc fills a descriptor that points into
 * the handle (valid until hlala_graph_file_free); hlala_loader_last_error holds the text of the last failure.
 * ---------------------------------------------------------------------------------------- */
typedef struct hlala_graph_file hlala_graph_file;
int  hlala_graph_load_text(const char* path, hlala_graph_file** out);
int  hlala_graph_cache_save(const hlala_graph_desc* graph, const char* path);
int  hlala_graph_cache_load(const char* path, hlala_graph_file** out);
int  hlala_graph_file_desc(const hlala_graph_file* g, hlala_graph_desc* desc);
void hlala_graph_file_free(hlala_graph_file* g);
const char* hlala_loader_last_error(void);

/* ------------------------------------------------------------------------------------------
 * BAM reader + seed extraction (host code, zlib): processBAM::extractSeeds2 (mapper/processBAM.cpp:703-864) with
 * protoSeeds::takeAlignment / isComplete (mapper/reads/protoSeeds.cpp:23-36, 371-380), sortChainsInSeeds (:1945-1967),
 * getAlignmentScore (:4314-4334).  A record is used if it is mapped (long-read mode: and not secondary), its reference carries
 * intervals, it has CIGAR operations and both its start and its end lie inside an interval.  Complete units (pairs with a primary
 * alignment for both mates; long reads: a primary alignment) come out in read-name order with their alignments sorted by the AS
 * tag (std::sort + std::reverse as in the reference), positions re-based to the interval start (chain_offset = 0, the contig of a
 * chain is the interval's `contig`).  long_read_mode != 0 yields the layout of hlala_batch_create_unpaired.
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    const char* ref_name;       /* BAM reference name (sequences.txt column Chr)                     */
    int32_t start_0based;       /* Start_1based - 1                                                   */
    int32_t stop_0based;        /* Stop_1based - 1                                                    */
    int32_t contig;             /* index into hlala_contigs_desc of the sequence this interval is     */
} hlala_bam_interval;
typedef struct hlala_seed_batch hlala_seed_batch;
int  hlala_bam_extract_seeds(const char* bam_path, int32_t n_intervals, const hlala_bam_interval* intervals, int32_t long_read_mode,
                             hlala_seed_batch** out);
/* The same with the number of decoding threads stated (0 = one per hardware thread, at most 32: measured on a 256-thread host the phases -- passes over
 * ~12 GB of inflated records bound by memory latency -- are fastest there; it is also one sample's share of such a host in BASELINE config 4).  The call returns
 * once the units, their order and every offset are known; names, bases, qualities, alignments and CIGARs of a unit are FILLED IN when a window that holds it is
 * first handed out (hlala_seed_batch_window, _desc -- the whole sample --, _name), on the same number of threads: a caller that walks the sample window by
 * window fills window i + 1 while the GPU aligns window i (HLALA_BAM_EAGER=1: everything at once, as before round 4).  BGZF blocks are independent gzip
 * members: they are inflated, their records parsed and grouped by read name in parallel; one final sort puts the complete units into
 * read-name order (the reference's std::map order, mapper/processBAM.cpp:712, 2024-2039).  The result does not depend on the thread count. */
int  hlala_bam_extract_seeds_mt(const char* bam_path, int32_t n_intervals, const hlala_bam_interval* intervals, int32_t long_read_mode,
                                int32_t n_threads, hlala_seed_batch** out);
/* ... with options.  HLALA_SEEDS_PACKED: the bases of the sample stay 4-bit packed as the BAM records hold them (a copy instead of an unpacking pass on the host, half the
 * bytes to upload): the descriptors of the sample carry read_bases_packed / first_read and a NULL read_bases (hlala_batch_in). */
#define HLALA_SEEDS_PACKED 1
int  hlala_bam_extract_seeds_opt(const char* bam_path, int32_t n_intervals, const hlala_bam_interval* intervals, int32_t long_read_mode,
                                 int32_t n_threads, int32_t flags, hlala_seed_batch** out);
/* The WHOLE sample as one descriptor (64-bit offsets starting at 0) pointing into the handle; counts[3] = records examined, seeds (read
 * names), incomplete seeds.  A sample beyond the size of one batch goes through the GPU window by window: hlala_seed_batch_window. */
int  hlala_seed_batch_desc(const hlala_seed_batch* s, hlala_batch_in* in, int64_t* counts);
/* the counters alone (nothing is filled in) */
int  hlala_seed_batch_counts(const hlala_seed_batch* s, int64_t* counts);
/* Units [first_unit, first_unit + n_units) of the sample as a batch descriptor (no copy: the window convention of hlala_batch_in).
 * HLALA_E_ARG outside the sample, HLALA_E_CAPACITY when the window itself exceeds the size of one batch. */
int  hlala_seed_batch_window(const hlala_seed_batch* s, int64_t first_unit, int32_t n_units, hlala_batch_in* in);
int64_t     hlala_seed_batch_units(const hlala_seed_batch* s);
const char* hlala_seed_batch_name(const hlala_seed_batch* s, int64_t unit);
/* decoding record: seconds[0..5] = block index, inflate, parse, group, name sort, layout (wall clock of each phase); threads used */
int  hlala_seed_batch_timing(const hlala_seed_batch* s, double* seconds6, int32_t* n_threads);
void hlala_seed_batch_free(hlala_seed_batch* s);
const char* hlala_bam_last_error(void);
/* "libdeflate" or "zlib": what inflates the BGZF blocks in this process (libdeflate's shared library if the machine has it, looked up at run time; HLALA_BAM_ZLIB=1
 * keeps zlib).  The decoded sample does not depend on it. */
const char* hlala_bam_inflate_engine(void);

/* Page-locked host memory for the buffers a caller hands to hlala_batch_create / hlala_batch_get_pairs_packed / the getters: transfers from and
 * to such buffers are true DMA (asynchronous, full PCIe rate); pageable buffers work everywhere, at about a third of the rate.  hlala_host_register
 * pins an existing allocation in place (e.g. the arrays of a seed batch: hlala_seed_batch_pin), hlala_host_unregister undoes it.  */
void* hlala_pinned_alloc(size_t bytes);
void  hlala_pinned_free(void* p);
int   hlala_host_register(void* p, size_t bytes);
int   hlala_host_unregister(void* p);
/* pins or unpins (pin = 0) the bulk arrays of a seed batch: read bases, qualities, chain records, CIGARs.  pin = 1: everything now (touches every page
 * of a sample whose windows are still unfilled); pin = 2: window by window -- hlala_seed_batch_window locks what the units up to the end of the window it
 * hands out occupy (64 MB granules, after filling them), so that a caller who walks the sample in ascending windows pays for the locking beside the GPU's
 * work on the batch before, not up front.  hlala_seed_batch_free unpins. */
int   hlala_seed_batch_pin(hlala_seed_batch* s, int pin);

/* ------------------------------------------------------------------------------------------
 * Insert-size estimation (processBAM::estimateInsertSize, mapper/processBAM.cpp:1071-1165, and
 * calculateInsertSizeFromHistogram :991-1069) on the kernels of stages A and B: for every pair of `in` the PRIMARY
 * alignment of mate 1 and of mate 2 (read_primary) is projected and extended; if the strands are valid every
 * distance of alignedReadPair_pairsDistancesUnderlyingSequences gets the weight 1/#distances in a histogram
 * (accumulated in pair order, as the reference walks its sorted read IDs); mean = weighted median,
 * sd = max(|median - 20 %|, |median - 80 %|).  The caller passes the first 4000 complete pairs (extractSeeds(4000), :1075).
 * A pair with a flagged chain (status < 0) is skipped and counted in n_skipped.  insert_mean / insert_sd of the context are
 * not used (and may be placeholders).                                                                                    */
typedef struct {
    double  mean, sd;                 /* calculateInsertSizeFromHistogram: weighted median / max deviation of the 20 % and 80 % points */
    int32_t n_used, n_skipped;        /* used_proto_seeds / skipped_proto_seeds (:1162)                                                */
    double  total_weight;             /* IS_total_size                                                                                 */
} hlala_insert_size_out;
int  hlala_estimate_insert_size(hlala_ctx* ctx, const hlala_batch_in* in, hlala_insert_size_out* out);
/* Replace insert_mean / insert_sd of a live context (the tables of the pairing step are rebuilt; graph and slabs stay): the context that estimated
 * the insert size goes on to align the sample (processBAM: estimateInsertSize, then IS_mean / IS_sd are set, mapper/processBAM.cpp:1071-1165).
 * Batches aligned before the call keep their results; call it with no alignment in flight. */
int  hlala_set_insert_size(hlala_ctx* ctx, double insert_mean, double insert_sd);

/* ------------------------------------------------------------------------------------------
 * Per-pair post-processing of alignReads_postSeedExtraction_andStoreInto (mapper/processBAM.cpp:2411-2446).
 *   coverage  : bases_per_level[level]++ for every column of both selected chains whose level is defined and whose graph
 *               character is not '_' (:2411-2428); the counters live in the context and accumulate over batches
 *               (they end up in reads_per_level.txt, processBAM.cpp:1902-1913)
 *   includeInHLA : a mate with a defined first level overlaps one of the gene intervals
 *               (HLATyper::intervalOverlapsWithGenes, hla/HLATyper.cpp:259-268: closed intervals, stop >= first && start <= last;
 *               the intervals are graphgene_levelBoundaries, HLATyper.cpp:241-252)
 * Pairs whose status is not 0 contribute nothing and are not included.
 * ---------------------------------------------------------------------------------------- */
int  hlala_set_gene_intervals(hlala_ctx* ctx, int32_t n_genes, const int32_t* first_level, const int32_t* last_level);
int  hlala_postprocess_pairs(hlala_ctx* ctx, hlala_batch* b, uint8_t* include_in_hla /* [n_pairs], may be NULL */);
/* bases_per_level[0 .. n_levels-2]; reset != 0 clears the counters afterwards */
int  hlala_get_coverage(hlala_ctx* ctx, int32_t* bases_per_level, int reset);

/* ------------------------------------------------------------------------------------------
 * HLATyper per-read log-likelihood scoring (hla/HLATyper.cpp).  One locus at a time.
 * Reads are the per-read lists of hla::oneExonPosition (hla/oneExonPosition.h:16-46) that survive the
 * HOST-side filters (mapQ_position >= 0.7, ignored alleles, ignored reads: HLATyper.cpp:2102-2121 --
 * none of them depends on the cluster, so the host folds them into `pos_use`).
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    int32_t        n_clusters;     /* C: allele clusters of the locus (HLAtype_clusters)                  */
    int32_t        exon_length;    /* P: length of the concatenated exon string                           */
    const uint8_t* cluster_seq;    /* [C*P] cluster_2_sequence, row-major                                 */
    int32_t        n_reads;        /* R: exonPositions_fromReads.size()                                   */
    const int32_t* pos_off;        /* [R+1] positions of read r: [pos_off[r], pos_off[r+1])               */
    const int32_t* pos_exon;       /* oneExonPosition::positionInExon                                     */
    const uint8_t* pos_g0;         /* first character of oneExonPosition::genotype                        */
    const int32_t* pos_glen;       /* genotype.length() (>1: inserted bases appended, HLATyper.cpp:3323)  */
    const uint8_t* pos_qual;       /* qualities.at(0)                                                     */
    const uint8_t* pos_use;        /* 0: skipped by the host-side filters                                 */
} hlala_exon_in;

/* likelihoods_perCluster_perRead / mismatches_perCluster_perRead (hla/HLATyper.cpp:2067-2277): LL[c*R + r], mism[c*R + r] */
int  hlala_exon_loglik(hlala_ctx* ctx, const hlala_exon_in* in, double* LL, int32_t* mism);
/* all cluster pairs c1 <= c2 in the reference's single-thread order (hla/HLATyper.cpp:2293-2364; Utilities::logAvg,
 * Utilities.cpp:1368-1379): pair (c1,c2) at index c1*C - c1*(c1-1)/2 + (c2-c1).  Outputs hold C*(C+1)/2 values. */
int  hlala_pair_loglik(hlala_ctx* ctx, const double* LL, const int32_t* mism, int32_t C, int32_t R,
                       double* pairLL, double* misAvg, double* misMin);

/* Exon positions of the read pairs of a batch for one locus (hla/HLATyper.cpp:1385-1428 with
 * oneReadAlignment_2_exonPositions_paired :3192-3565, alignmentFractionOK :3082-3101, alignmentWeightedOKFraction :3933-4018,
 * alignerBase::alignedReadPair_pairsDistanceInGraphLevels alignerBase.cpp:246-283, removeDoublePositionsFromRead :4020-4083).
 * A pair yields one entry of exonPositions_fromReads when it passes the pair test (:1404-1410: strands valid, |distance - insert mean|
 * <= 5 sd, mapQ of mate 1 >= min_mapq, both weighted-OK fractions >= min_weighted_ok) and at least one of its columns lies on an exon
 * level; positions come out as the reference's std::map order (ascending graph level, one per level: the alternative with the best
 * worst-quality, first wins).  Reads are taken in alignment orientation (read_bases / read_quals of the batch), which is the base the
 * reference reaches through its reverse-index arithmetic (:3307-3321).
 * On an UNPAIRED batch (hlala_batch_create_unpaired) the unit is a read: oneReadAlignment_2_exonPositions_unpaired (:3568-3930) and the
 * test of :1476 (mapQ >= min_mapq and at least min_alignment_columns alignment columns); the "mate 2" halves of the per-read arrays
 * hold -1 (pairedRead_* = -1, :3574-3580) and read_distance is -1.                                                                */
typedef struct {
    int32_t        level_min, level_max; /* combined_exon_sequences_graphLevels_min / _max                                  */
    const int32_t* level_to_exon;        /* [level_max - level_min + 1] graphLevel_2_exonPosition; -1 = not an exon level   */
    double         insert_mean, insert_sd;
    double         min_mapq;             /* minimumMappingQuality (0.0, HLATyper.cpp:30)                                    */
    double         min_weighted_ok;      /* min_bothReads_weightedCharactersOK (0.0, HLATyper.cpp:28)                       */
    const uint8_t* pair_mask;            /* [n_pairs] or NULL: only pairs with a non-zero entry are looked at (includeInHLA) */
    int32_t        min_alignment_columns;/* unpaired batches only: minAlignmentLength_unpaired (1000, HLATyper.cpp:1032, test at :1476) */
    int32_t        reserved;
} hlala_locus_desc;

typedef struct {
    int32_t  cap_reads, cap_pos, cap_chars;  /* in: capacities of the arrays below                                          */
    int32_t  n_reads, n_pos, n_chars;        /* out: exonPositions_fromReads.size(), total positions, total genotype chars  */
    int32_t  n_pairs_ok, n_pairs_broken;     /* out: readPairs_OK / readPairs_broken (:1424, :1462), over the looked-at pairs */
    int32_t* read_pair;        /* [cap_reads] pair index in the batch                                                        */
    double*  read_weighted_ok; /* [2*cap_reads] alignmentWeightedOKFraction of mate 1, mate 2                                 */
    double*  read_fraction_ok; /* [2*cap_reads] alignmentFractionOK of mate 1, mate 2                                         */
    int32_t* read_distance;    /* [cap_reads] alignedReadPair_pairsDistanceInGraphLevels                                      */
    int32_t* read_cols_nongap; /* [2*cap_reads] alignmentColumnsWithAtLeastOneNonGap of mate 1, mate 2                        */
    int32_t* pos_off;          /* [cap_reads+1] positions of read i: [pos_off[i], pos_off[i+1])                               */
    int32_t* pos_exon;         /* oneExonPosition::positionInExon                                                            */
    int32_t* pos_level;        /* ::graphLevel                                                                               */
    uint8_t* pos_mate;         /* 1 / 2: the mate the position comes from (fromFirstRead)                                    */
    uint8_t* pos_mapq;         /* Phred character of mapQ_perPosition; ::mapQ_position = PhredToPCorrect of it               */
    int32_t* pos_novel_gap;    /* ::runningNovelGapEitherDirection                                                           */
    int32_t* geno_off;         /* [cap_pos+1] genotype / qualities of position j: [geno_off[j], geno_off[j+1])               */
    uint8_t* geno_chars;       /* ::genotype                                                                                 */
    uint8_t* qual_chars;       /* ::qualities (0 where the genotype is "_" and carries no quality)                            */
    uint8_t* read_reverse;     /* [2*cap_reads] ::reverse of the positions of mate 1, mate 2 (strand of the chosen alignment); may be NULL */
    double*  read_mapq;        /* [2*cap_reads] ::mapQ = ::mapQ_genomic of the positions of mate 1, mate 2 (-1: no mate 2); may be NULL     */
} hlala_exon_positions_out;

/* returns HLALA_E_CAPACITY (with the needed n_* filled in) when an array is too small */
int  hlala_exon_positions(hlala_ctx* ctx, hlala_batch* b, const hlala_locus_desc* locus, hlala_exon_positions_out* out);

/* Read / allele filters between the exon positions and the likelihoods (hla/HLATyper.cpp:1496-1720 filterFirst20, :1722-1862
 * high-coverage allele filter, and the use test of the likelihood loop :2102-2120).  pos_use[j] = 1 iff position j enters the
 * likelihood: mapQ_position >= min_per_position_mapq, its read is not ignored, its allele is not ignored at its exon position.
 * This step runs on the HOST: which reads make the "first 20" of a position is decided by std::sort on tied keys (:1557-1565), i.e.
 * by the C++ library's unspecified tie order; calling the same library routine on the same sequence is the only way to agree with
 * the reference, and the data is a few hundred thousand small records per locus.  No context is needed.                          */
typedef struct {
    int32_t filter_first20;              /* filterFirst20 (true, HLATyper.cpp:73); also the divisor of first20_prop, :1593 (sic)   */
    int32_t first20_n;                   /* filterFirst20N (20)                                                                     */
    double  first20_min_prop;            /* filterFirst20MinProp (0.1)                                                              */
    int32_t first20_limit_per_read;      /* filterFirst20MinProp_limitKickOutPerRead (2, HLATyper.h:57)                              */
    double  min_per_position_mapq;       /* minimumPerPositionMappingQuality (0.7, HLATyper.cpp:31)                                  */
    int32_t high_coverage_filter;        /* highCoverage_filter_alleles (false; true with min coverage 1 / freq 0.15 at :944-946)    */
    int32_t high_coverage_min_coverage;  /* highCoverage_minCoverage (100)                                                          */
    double  high_coverage_min_freq;      /* highCoverage_minAlleleFreq (0.2)                                                        */
    int32_t long_read_strand_filter;     /* longReadsMode set && longReads_filterStrand (true, HLATyper.cpp:77): alleles seen >=      */
    int32_t strand_min_allele_coverage;  /* longReads_filterStrand_minAlleleCoverage (100) times whose rarer strand is below         */
    double  strand_min_freq;             /* longReads_filterStrand_minStrandFreq (0.1) are ignored, :1846-1855; needs read_reverse    */
} hlala_filter_params;
typedef struct {
    int64_t considered_positions, positions_with_removed_alleles, considered_alleles, removed_alleles;   /* :1703-1707 */
    int64_t reads_kicked_out, reads_kicked_out_robust;                                                   /* :1709, :1714 */
    int64_t high_coverage_positions, high_coverage_removed_alleles;                                      /* :1865-1868 */
    int64_t bases_used;                                                                                  /* HLATypeInference_thisLocus_bases_used, :2122 */
    int64_t strand_alleles_enough_coverage, strand_removed_alleles, strand_positions_with_removed;      /* :1874-1876 (the last one counts as the reference does: once per allele visited after a removal) */
} hlala_filter_stats;
int  hlala_filter_positions(const hlala_exon_positions_out* pos, const hlala_filter_params* prm, uint8_t* pos_use /* [n_pos] */,
                            uint8_t* read_ignored /* [n_reads] or NULL */, hlala_filter_stats* stats /* or NULL */);

/* The call of one locus from the all-pairs table (hla/HLATyper.cpp:2366-2541).  Pair (c1 <= c2) sits at the index
 * hlala_pair_loglik uses.  order = pair indices sorted by LL descending, Mism_avg ascending (std::sort + std::reverse, :2381-2403;
 * the order among pairs equal in both keys is unspecified in the reference too -- n_sort_ties counts adjacent equal keys);
 * p_normalized = exp(LL - max) / sum (:2411-2448); cluster_marginal = clusterI_overAllPairs, accumulated in `order` (:2459-2486);
 * first = first maximum of the marginals in cluster order (findIntMapMax, Utilities.cpp:257-272, :2490); second = among the pairs that
 * contain `first` the maximum P, ties resolved by the smallest Mism_min, then the smallest cluster (:2498-2533). */
typedef struct {
    int32_t first_cluster;     /* bestGuess_firstAllele.second                    */
    int32_t second_cluster;    /* bestGuess_secondAllele.second                   */
    double  first_marginal;    /* bestGuess_firstAllele.first                     */
    double  second_p;          /* oneBestGuess_secondAllele.first                 */
    double  ll_max;            /* findVectorMax(LLs_completeReads).first (:2408)  */
    int32_t max_pair;          /* ... .second (first maximum)                     */
    int32_t n_sort_ties;
} hlala_call_out;
int  hlala_call_locus(hlala_ctx* ctx, int32_t C, const double* pairLL, const double* misAvg, const double* misMin,
                      int32_t* order /* [C(C+1)/2] or NULL */, double* p_normalized /* [C(C+1)/2] or NULL */,
                      double* cluster_marginal /* [C] or NULL */, hlala_call_out* out);

/* The three steps of a locus in one call -- hlala_exon_loglik, hlala_pair_loglik, hlala_call_locus (HLATyper.cpp:2067-2541) -- with the tables left on the
 * device in between: the per-read table (clusters x reads; LL / mism may be NULL: not downloaded) is not moved at all, the all-pairs tables come down once and
 * are not uploaded again for the call.  Every output equals, bit for bit, what the three calls return one after the other. */
int  hlala_type_locus(hlala_ctx* ctx, const hlala_exon_in* in, double* LL /* [C*R] or NULL */, int32_t* mism /* [C*R] or NULL */,
                      double* pairLL, double* misAvg, double* misMin /* [C(C+1)/2] each */, int32_t* order, double* p_normalized, double* cluster_marginal, hlala_call_out* out);

/* Reference contigs of a graph directory (host code): one contig per row of <graph_dir>/sequences.txt -- the stretch
 * [Start_1based, Stop_1based] of the BAM reference it names (column Chr, or PRG_<SequenceID>; extended_reference_genome == 0: the whole
 * sequence of mapping_PRGonly/referenceGenome.fa) with the levels of translation/<SequenceID>.txt (processBAM::initBAM
 * mapper/processBAM.cpp:1183-1402, the constructor :69-88, _loadMapping :4389-4457).  hlala_contigs_file_desc is the `contigs` argument
 * of hlala_create, hlala_contigs_file_intervals the interval list of hlala_bam_extract_seeds (interval i = contig i).  A translation
 * file that ends in a newline yields one extra level-0 entry in the reference's table; the contig then carries one extra position
 * (base 'N', level 0) so that the level -> (sequence, position) table comes out the same. */
typedef struct hlala_contigs_file hlala_contigs_file;
int     hlala_contigs_load_dir(const char* graph_dir, int32_t extended_reference_genome, hlala_contigs_file** out);   /* errors: hlala_loader_last_error */
/* The same in two steps, for a caller that wants the BAM decoder running while the bulk of the directory is still being read: hlala_contigs_open_dir reads
 * sequences.txt and the reference sequences it names -- enough for hlala_contigs_file_intervals --, hlala_contigs_load_translations the translation tables
 * (tens of millions of lines; parsed side by side) that hlala_contigs_file_desc / hlala_create need.  hlala_contigs_load_dir = both. */
int     hlala_contigs_open_dir(const char* graph_dir, int32_t extended_reference_genome, hlala_contigs_file** out);
int     hlala_contigs_load_translations(hlala_contigs_file* c);
int     hlala_contigs_file_desc(const hlala_contigs_file* c, hlala_contigs_desc* desc);
int32_t hlala_contigs_file_intervals(const hlala_contigs_file* c, hlala_bam_interval* out, int32_t cap);              /* returns the number of intervals */
void    hlala_contigs_file_free(hlala_contigs_file* c);

/* ------------------------------------------------------------------------------------------
 * HLATyper host side around the kernels (host code; hla/HLATyper.cpp).  hlala_typer_open reads <graph_dir>/PRG/segments.txt and the
 * first line of every segment file: graph level names (Graph::readGraphLoci, Graph/Graph.cpp:2563-2614 -> graphLocus_2_levels,
 * HLATyper.cpp:84-92) and the level range of every gene (graphgene_levelBoundaries, :104-214: the intervals of
 * hlala_set_gene_intervals).  hlala_typer_locus reads the exon files of one locus (find_file_for_exon :3130-3200; exon_ids NULL =
 * the table of fill_loci_2_exons :2812-2846) and clusters alleles with identical exon sequences (:1180-1372): cluster_seq /
 * level_min / level_max / level_to_exon are the inputs of hlala_exon_positions and hlala_exon_loglik.
 * ---------------------------------------------------------------------------------------- */
typedef struct hlala_typer hlala_typer;
typedef struct hlala_locus hlala_locus;
int  hlala_typer_open(const char* graph_dir, hlala_typer** out);
void hlala_typer_close(hlala_typer* t);
const char* hlala_typer_last_error(void);
int32_t     hlala_typer_n_levels(const hlala_typer* t);
const char* hlala_typer_level_name(const hlala_typer* t, int32_t level);
int32_t     hlala_typer_level_of(const hlala_typer* t, const char* graph_locus_id);              /* -1: unknown */
int32_t     hlala_typer_n_genes(const hlala_typer* t);                                            /* genes in name order (std::map) */
int  hlala_typer_gene(const hlala_typer* t, int32_t i, const char** name, int32_t* first_level, int32_t* last_level);
/* G groups (hla_nom_g.txt; read_G_alleles / translate_allele_list_to_G_allele, HLATyper.cpp:4086-4207) for R1_bestguess_G.txt */
int  hlala_typer_load_g_groups(hlala_typer* t, const char* hla_nom_g_path);
/* translate_allele_list_to_G_allele (hla/HLATyper.cpp:4095-4148) for a ';'-joined allele list (e.g. hlala_locus_cluster_id): the G group all
 * known members share (*perfectly = 1), the most frequent one (0), or the list itself when no member is in the table (0).  HLALA_E_STATE
 * when no table is loaded or the locus is not in it (can_translateToG_locus, :4086-4092).                                            */
int  hlala_typer_g_translate(const hlala_typer* t, const char* alleles, char* out, int32_t cap, int32_t* perfectly);
int  hlala_typer_locus(const hlala_typer* t, const char* locus, int32_t n_exons, const char* const* exon_ids, hlala_locus** out);
void hlala_locus_free(hlala_locus* l);
typedef struct {
    int32_t n_clusters, n_columns, n_exons, level_min, level_max, n_types;
    const uint8_t* cluster_seq;       /* [n_clusters * n_columns] cluster_2_sequence                                        */
    const int32_t* level_to_exon;     /* [level_max - level_min + 1] graphLevel_2_exonPosition, -1 = not an exon level      */
    const int32_t* col_level;         /* [n_columns] combined_exon_sequences_graphLevels                                    */
    const int32_t* col_exon;          /* [n_columns] ..._individualExon                                                     */
    const int32_t* col_exon_pos;      /* [n_columns] ..._individualExonPosition                                             */
    const int32_t* exon_length;       /* [n_exons] exon_lengths                                                             */
} hlala_locus_info;
int  hlala_locus_get(const hlala_locus* l, hlala_locus_info* out);                               /* pointers into the handle */
const char* hlala_locus_cluster_id(const hlala_locus* l, int32_t cluster);                       /* members joined by ";" (std::set order) */
int32_t     hlala_locus_type_cluster(const hlala_locus* l, const char* hla_type);                /* HLAtype_2_clusterID, -1: unknown */
/* k-mers of a cluster's exon sequences (gaps removed, exon by exon): n_total counts all, the ones without '*' go to queries
 * (k characters each, no terminator) -- the query set of hlala_kmer_presence (calculcatekMerPresence, HLATyper.cpp:2652-2688) */
int  hlala_locus_cluster_kmers(const hlala_locus* l, int32_t cluster, int32_t k, char* queries, int32_t cap_queries, int32_t* n_queries, int32_t* n_total);

/* Which of the query k-mers occur in the reads of a batch (the k-mer index of HLATypeInference, HLATyper.cpp:999-1027, asked the way
 * :2652-2688 asks it): present[q] = 1 iff some read of a looked-at unit holds a k-mer whose canonical form
 * (kMer_canonical_representation :4237-4256: the smaller of the k-mer and its reverse complement) equals the canonical form of query q.
 * Runs on the GPU over the reads resident in the batch: no index is built, every read k-mer is looked up in the sorted query set.
 * k <= 31, queries over ACGT (others never match: the reads' N never equals an allele character); pair_mask as in hlala_locus_desc. */
int  hlala_kmer_presence(hlala_ctx* ctx, hlala_batch* b, const uint8_t* pair_mask, int32_t k, int32_t n_queries, const char* queries, uint8_t* present);
/* The same questions asked of reads that were KEPT on the device while their batches were resident: the reference builds its k-mer index while it walks
 * the reads (HLATyper.cpp:999-1027) and asks after the calls (:2652-2688); a sample that goes through the GPU in batches has released most of them by
 * then.  hlala_kmer_keep_reads appends the bases of the looked-at units of `b` (pair_mask as above; null: all) to the context's store -- a device-to-device
 * copy of the few per cent of the reads that overlap the loci -- and hlala_kmer_presence_kept answers for the union of everything kept since the last
 * hlala_kmer_forget_reads (also called by hlala_destroy).  present[q] equals the OR over the batches of what hlala_kmer_presence would return. */
int  hlala_kmer_keep_reads(hlala_ctx* ctx, hlala_batch* b, const uint8_t* pair_mask, int64_t* n_reads_kept /* may be null */);
int  hlala_kmer_presence_kept(hlala_ctx* ctx, int32_t k, int32_t n_queries, const char* queries, uint8_t* present);
void hlala_kmer_forget_reads(hlala_ctx* ctx);

/* Per-unit alignment statistics of a batch (the loop of hla/HLATyper.cpp:1043-1097 and the per-pair quantities of the pair test
 * :1404-1410): alignedReadPair_strandsValid, alignedReadPair_pairsDistanceInGraphLevels (alignerBase.cpp:246-283), alignmentFractionOK
 * (:3082-3101) and alignmentWeightedOKFraction (:3933-4018) of both mates, alignment columns (graph_aligned.size()) and the mates'
 * mapping qualities.  valid[u] = 0 for a unit whose chains hit a device capacity (everything else of that unit is 0).  Unpaired
 * batches fill the mate-1 halves; the mate-2 halves hold -1 and distance is -1. */
typedef struct {
    uint8_t* valid;          /* [n_units]                                   */
    uint8_t* strands_valid;  /* [n_units]                                   */
    int32_t* distance;       /* [n_units]                                   */
    double*  fraction_ok;    /* [2*n_units]                                 */
    double*  weighted_ok;    /* [2*n_units]                                 */
    int32_t* n_columns;      /* [2*n_units]                                 */
    double*  mate_mapq;      /* [2*n_units]                                 */
} hlala_unit_stats_out;
int  hlala_unit_alignment_stats(hlala_ctx* ctx, hlala_batch* b, hlala_unit_stats_out* out);

/* summaryStatistics.txt (hla/HLATyper.cpp:1030-1125) over the units with a non-zero unit_mask entry (NULL: all): the units of a paired
 * batch are the paired alignments, those of an unpaired batch the unpaired ones (the other group is empty, as in a run of the reference
 * on one kind of reads).  Sums run in unit order like the reference's loops. */
int  hlala_typer_write_summary(const char* out_dir, int32_t n_units, int32_t unpaired, const uint8_t* unit_mask, const hlala_unit_stats_out* stats,
                               double insert_mean, double insert_sd, int32_t min_alignment_length_unpaired);

/* Result files of one locus (HLATyper.cpp:1883-2044 pile-up + read IDs, :2451-2488 all pairs, :2543-2759 coverage, column
 * incompatibilities, best guesses).  hlala_typer_begin_output creates the directory and the headers of R1_bestguess.txt /
 * R1_bestguess_G.txt, hlala_locus_write_files writes R1_pileup_<locus>.txt, R1_readIDs_<locus>.txt, R1_PP_<locus>_pairs.txt,
 * R1_columnIncompatibilities_<locus>.txt and appends the two best-guess rows (the G rows if hlala_typer_load_g_groups knows the
 * locus), hlala_typer_end_output writes R1_parameters.txt.  Numbers are printed by the same iostream calls as the reference.
 * With unit_stats given, the lines of histogram_matchesPerRead.txt of the locus are appended too (header: hlala_typer_begin_output):
 * "read" / "readPair" per pair that passes the pair test of the locus (:1404-1429; paired batches only) and "base" per piled position (:1928). */
typedef struct {
    const hlala_exon_positions_out* pos;     /* hlala_exon_positions of this locus, with read_reverse and read_mapq             */
    const hlala_filter_params* filter;       /* the parameters hlala_filter_positions ran with                                    */
    const char* const* unit_name_1;          /* [units of the batch] read.name of mate 1 (indexed by pos->read_pair[])            */
    const char* const* unit_name_2;          /* ... of mate 2; NULL for unpaired batches (pairedRead_ID = "")                     */
    int32_t long_read_mode;                  /* longReadsMode set: positions with runningNovelGapEitherDirection >= 2 are not piled */
    int32_t n_clusters;
    const double*  pair_ll;                  /* hlala_pair_loglik                                                                  */
    const double*  mis_avg;
    const double*  mis_min;
    const int32_t* order;                    /* hlala_call_locus                                                                   */
    const double*  p_normalized;
    const hlala_call_out* call;
    double  kmers_covered[2];                /* proportionkMersCovered of the first / second called allele (-1: no k-mers)         */
    int32_t unaccounted_min_coverage;        /* threshold_reportColumn_forPresenceOfUnaccountedAlleles_minCoverage (30, :67)        */
    int32_t pairs_file_done;                 /* 1: R1_PP_<locus>_pairs.txt is already there (hlala_locus_write_pairs_file); 0: write it  */
    double  unaccounted_min_fraction;        /* ..._minAlleleFraction (0.2, :68)                                                   */
    const hlala_unit_stats_out* unit_stats;  /* NULL: no histogram lines                                                           */
    const uint8_t* unit_mask;                /* [units of the batch] or NULL: the mask hlala_exon_positions ran with (includeInHLA) */
    int32_t n_units;                         /* units of the batch                                                                 */
    int32_t reserved2;
    double  insert_mean, insert_sd, min_mapq, min_weighted_ok;   /* the pair test of hlala_locus_desc                            */
} hlala_locus_report_in;
typedef struct {
    double  locus_coverage, first_decile_coverage, minimum_coverage, avg_column_error, min_column_p;
    int64_t bases_used;
    int32_t n_columns_unaccounted, n_utilized_reads, n_piled_positions, reserved;
} hlala_locus_report_out;
int  hlala_typer_begin_output(const char* out_dir, double unaccounted_min_fraction);
int  hlala_locus_write_files(const hlala_locus* l, const hlala_locus_report_in* in, const char* out_dir, hlala_locus_report_out* out);
/* The all-pairs table R1_PP_<locus>_pairs.txt alone (HLATyper.cpp:2451-2488; millions of lines for a class-I locus): what it prints is known as soon as
 * hlala_call_locus returns, so a host program can write it beside the k-mer pass and the typing of the next locus and set pairs_file_done above. */
int  hlala_locus_write_pairs_file(const hlala_locus* l, int32_t n_clusters, const int32_t* order, const double* p_normalized, const double* pair_ll,
                                  const double* mis_avg, const char* out_dir);
int  hlala_typer_end_output(const char* out_dir, const char* loci_comma_separated, int32_t very_conservative_read_likelihoods);

/* Known-answer helpers exported for the parity tests (device implementations of
 * Utilities::PCorrectToPhred / PhredToPCorrect, Utilities.cpp:178-203, 357-377, and of glibc
 * rand_r as used by Utilities::randomNumber_nonCritical, Utilities.cpp:922).                 */
int  hlala_kat_phred(hlala_ctx* ctx, int n, const double* p_correct, uint8_t* phred_out,
                     const uint8_t* phred_in, double* p_out);
int  hlala_kat_rand_r(hlala_ctx* ctx, int n, uint32_t* seeds_inout, int32_t* values_out);
/* the exponential of the posteriors of the pairing step (arguments <= 0): correctly rounded on the device, see device_common.h */
int  hlala_kat_exp(hlala_ctx* ctx, int n, const double* x, double* exp_x);

/* sizeof() of the structs of this header as the library was compiled, by struct name ("hlala_graph_desc", "hlala_params", ...);
 * -1 for an unknown name.  Lets a foreign-function binding (ctypes, cgo, JNI) check its mirror of the layout at load time. */
int  hlala_abi_sizeof(const char* struct_name);

/* Version of this interface.  It changes whenever the meaning or the type of a field changes WITHOUT changing the size of its struct (which
 * hlala_abi_sizeof cannot see) or a struct grows: 2 = hlala_batch_in carries 64-bit window offsets and an absolute read_primary (round 3);
 * 3 = hlala_batch_stats ends with n_dp_jump_free / ms_dp_jump_free, hlala_batch_in with read_bases_packed / first_read (round 4);
 * 4 = hlala_batch_stats ends with n_dp_band / n_dp_band_failed / n_dp_jump_free_failed / ms_dp_band (round 5);
 * 5 = ... with n_dp_band2 / n_dp_band2_failed / ms_dp_band2, hlala_set_tail_pool / hlala_flush / hlala_comm_* exist (round 6).  A caller compares
 * hlala_abi_version() with the HLALA_ABI_VERSION it was compiled against and refuses to run on a mismatch (hla-la_amd/__init__.py and
 * hla-la_amd/host/hlala_host.hpp do). */
#define HLALA_ABI_VERSION 5
int  hlala_abi_version(void);
/* ---- debug / diagnostics section (tests and tools only; not part of the path the reference calls).  Layouts may change between rounds: the constants below are
 * checked against the library's own at compile time (hlala_api.hip). */
#define HLALA_DEBUG_WC_N           88     /* work counters of a batch (csrc/batch.h: B.work_counter)                                                   */
#define HLALA_DEBUG_WC_BAND_FETCH  48     /* [+0..5] items fetched by the 16 / 32 / 64-lane band kernels, left and right                               */
#define HLALA_DEBUG_WC_BAND_WHY    62     /* [+2..5] band calls that failed over: past the staged levels, past the linear run, too many iterations, ties */
#define HLALA_DEBUG_WC_BAND_TIED   68     /* band calls whose end cell was drawn among equal sequence-complete cells                                   */
/* the first min(capacity, HLALA_DEBUG_WC_N) work counters of the batch's last stages */
int  hlala_debug_work_counters(hlala_ctx* ctx, hlala_batch* batch, int* out, int capacity);
/* the DP items of the batch's last extension stage ([2 * n_chains] records of 8 ints: item, read offset, read length, start offset, start level, start node, class,
 * linear run; item < 0: no call) and its retry / fail-over lists ([16 * n_chains] slots); either pointer may be NULL; capacities in bytes are checked */
int  hlala_debug_dp_items(hlala_ctx* ctx, hlala_batch* batch, int* items_out, long long items_capacity_bytes, int* retry_out, long long retry_capacity_bytes);
/* counters[32] of a batch (stage statistics, timing builds), the phase-clock buffer of profile builds (8192 ints), device memory of a batch / its context (4 values) */
int  hlala_debug_counters(hlala_ctx* ctx, hlala_batch* batch, unsigned long long* out32);
int  hlala_debug_buffer(hlala_ctx* ctx, int* out8192, int clear);
int  hlala_debug_memory(hlala_ctx* ctx, hlala_batch* batch, unsigned long long* out4);

/* bit mask of build-time variants compiled into this library: HLALA_BUILD_AGENT_RELEASE = the in-memory DP class releases at agent scope
 * (make EXTRA=-DHLALA_DP_AGENT_RELEASE: the fences of rounds 2-4, kept as a switch for a device in threadgroup-split mode, kernel_dp.hip dp_sync) */
#define HLALA_BUILD_AGENT_RELEASE 2
int  hlala_build_flags(void);

#ifdef __cplusplus
}
#endif
#endif /* HLALA_GPU_H_ */
