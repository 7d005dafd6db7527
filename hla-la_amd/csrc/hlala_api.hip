// hlala_api.hip -- the C ABI of include/hlala_gpu.h: device memory management, host tables, launches.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <algorithm>
#include <map>
#include <mutex>
#include <chrono>
#include <thread>
#include <set>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/hlala_gpu.h"
#include "batch.h"
#include "flat_graph.hpp"
#include "host_internal.h"

// unity build: the kernels live in their own files but are compiled in this translation unit
#include "kernel_dp.hip"
#include "kernel_project.hip"
#include "kernel_order.hip"
#include "kernel_pair.hip"
#include "kernel_typer.hip"
#include "kernel_call.hip"
#include "kernel_exonpos.hip"
#include "kernel_kmer.hip"
#include "kernel_dp_band.hip"
#include "kernel_dp_band2.hip"

namespace hlala {
size_t proj_slab_bytes_host(int stride, int maxNodesPerLevel) { return proj_slab_bytes(stride, maxNodesPerLevel); }
}  // namespace hlala

using namespace hlala;

static thread_local std::string g_create_error;

struct DevBuf {
    void* p = nullptr; size_t bytes = 0;
};

struct hlala_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    // hlala_align_batch (fused, paired batches) runs the DP classes from DP_SIDE_TIER on -- few, long DP calls that leave most of the chip idle --
    // on this second, low-priority stream, then the second stitch / pairing pass over the pairs that waited for them; the main stream goes on with the
    // pairs they do not concern and, when the caller has more than one batch in flight, with the next batch
    hipStream_t side = nullptr;
    // `up` carries the uploads of hlala_batch_create, `rs` everything that only READS a batch or works on caller data (getters, post-processing, exon
    // positions, typer scoring): neither waits for alignments of OTHER batches queued on the main stream, so a caller with two batches in flight
    // uploads the next batch and fetches the previous one while the current one is being aligned.  `active` = the stream the helpers use right now.
    hipStream_t up = nullptr, rs = nullptr, active = nullptr;
    // tail pool (round 6, hlala_set_tail_pool): fused alignments leave their broad / large / in-memory DP classes pending; flush_tail runs them ONCE for the pooled batches
    int tail_pool_k = 1; std::vector<hlala_batch*> tail;
    hipEvent_t evSideTail = nullptr; bool sideTailValid = false;      // end of the last work queued on the side stream: a non-fused launch of the classes that share its slabs waits for it
    hlala_params params{};
    FlatGraph F;
    DevGraph G{};
    DevGraph* dG = nullptr;       // device copy of G (kernels take descriptors by pointer)
    DevTables* dT = nullptr;
    double* d_islog = nullptr;
    long long* d_contig_off = nullptr; uint8_t* d_contig_seq = nullptr; int* d_contig_level = nullptr;
    int n_contigs = 0; std::vector<long long> contig_off;
    std::vector<void*> allocs;
    // Device buffers of destroyed batches are kept for the next batch (hipMalloc of the ~35 GB of column arrays of a 1 M-pair batch
    // costs 0.6-1.1 s): block sizes by pointer, free blocks by size.
    std::unordered_map<void*, size_t> block_bytes;
    std::multimap<size_t, void*> pool;
    size_t pool_bytes = 0;
    size_t pool_cap = (size_t)176 << 30;      // bytes the pool may keep parked (HLALA_POOL_CAP_GB: several processes sharing one device, e.g. the N-rank dry runs on a one-GPU box)
    std::mutex pool_mu;              // the pool and block_bytes: this context's thread, and any thread that meets an out-of-memory error on the device (device_malloc_retry)
    std::set<struct hlala_batch*> batches;     // live batches: detached (not dangling) if the context is destroyed first
    // DP scratch slabs: one per DpTiny group (4 per wave), one per DpSmall / DpLarge wave (same pool, same layout size)
    int jf_margin = 16;      // (measured 4 / 8 / 16 / 48: 16-lane + 32-lane class 105.6 / 104.1 / 103.4 / 104.1 ms -- a tight bound sends more calls to the cheap instantiation and more of them on to the 32-lane class) levels beyond the read bases left that a jump-free call is taken to reach (kernel_dp.hip: k_dp_items)
    int stitch_by_row = 1;     // k_stitch_chains draws column rows (position order, only chains that hold one) instead of chain numbers (HLALA_STITCH_BY_ROW=0; 6.9 -> 6.65 ms)
    int stitch_draw = 12;      // chains per draw of k_stitch_chains (8: 9.1 ms per million pairs, the rate of the draws themselves; 12 / 16: 6.7 / 6.6 ms; 64: 15 ms -- kernel_dp.hip)
    bool side_after_pair = false; // HLALA_SIDE_AFTER_PAIR=1: the side-stream classes are queued behind the main stream's stitch and pairing passes instead of beside them (measured: the pairing pass 20.9 -> 4.1 ms, but the next batch's projection 35.5 -> 54.8 ms beside the wide class instead; step 183.4 -> 185.5 ms)
    bool rows_all = false;        // HLALA_ROWS_ALL=1: column rows for every chain of a batch, the filters run with the projection (rounds 1-4)
    bool band_risky = false;      // HLALA_DP_BAND_RISKY=1 (tests: force fail-overs of the band kernel)
    int band2_maxj = B2_MAXJ64;
    int band2_grid = 0, band2_margin = 12; u64* band2_slabs = nullptr;      // the two-track band kernels (kernel_dp_band2.hip): blocks (0: HLALA_DP_BAND2=0), levels beyond the read bases left that the track run must cover (HLALA_DP_BAND2_MARGIN), back-pointer slabs
    int band_grid = 0, band_margin = 8;      // the band kernel in front of the 16-lane class (kernel_dp_band.hip): blocks (0: HLALA_DP_BAND=0) and the levels beyond the read bases left a call is taken to reach (HLALA_DP_BAND_MARGIN)
    char* tiny_slabs = nullptr; size_t tiny_slab_bytes = 0; int tiny_grid = 0; int jf_grid = 0;      // jf_grid: blocks of the jump-free instantiation of the 16-lane class (0: not used)
    char* ext_slabs = nullptr; size_t ext_slab_bytes = 0; char* wide_slabs = nullptr; char* mid_slabs = nullptr; char* large_slabs = nullptr; size_t large_slab_bytes = 0; char* huge_slabs = nullptr; size_t huge_slab_bytes = 0; int huge_grid = 0; int ext_grid = 0; int wide_grid = 0; int broad_grid = 0; int retry_grid = 0; int stitch_grid = 0; int mid_grid = 0; size_t mid_slab_bytes = 0;
    char* proj_slabs = nullptr; size_t proj_slab_bytes = 0; int proj_grid = 0, pair_grid = 0, pair_lean_grid = 0;
    char* rethread_slabs = nullptr; size_t rethread_slab_bytes = 0; int rethread_grid = 0;      // k_rethread_chains: back pointers of one chain per wave (short reads; HLALA_RETHREAD=0 turns the kernel off)
    double* pair_scratch = nullptr;   // [2 * pair_grid][PAIR_COMB]: combination tables of the rare pairs with more than PAIR_COMB_LDS combinations (main- and side-stream pass)
    char* proj_long_slabs = nullptr; size_t proj_long_slab_bytes = 0;      // long reads only (max_columns > 512): column / window arrays of k_project_chains<ProjLdsLong>
    void* create_scratch = nullptr; size_t create_scratch_bytes = 0; bool create_scratch_pinned = false;      // host scratch of hlala_batch_create (page-locked when the runtime grants it; one caller thread per context)
    int proj_long_stagger = 0;              // long-read projection: wavefront w starts (w mod 64) x this many cycles after the kernel does (HLALA_PROJ_LONG_STAGGER; k_project_chains)
    int long_chunk_nodes = 1 << 20, long_max_segs = 1 << 20;      // (batch.h; the kernel clamps them to its array sizes)
    int order_cost = 0;                     // long-read layout: heaviest windows first (batch.h: order_cost; HLALA_LONG_ORDER=0: position order)
    int order_shift = 8, order_nb = 0;      // position buckets of a batch's chains (kernel_order.hip); order_nb 0: input order (HLALA_LOCALITY=0)
    int* dbg_host = nullptr;      // (device memory) non-null with HLALA_DEBUG=1: kernels accumulate phase clocks into the batch counters (hlala_debug_counters)
    // per-pair post-processing: coverage counters [L-1] and gene intervals
    int* d_cov = nullptr; int n_cov = 0; int* d_gene_first = nullptr; int* d_gene_last = nullptr; int n_genes = 0;
    // reads kept for the k-mer questions (hlala_kmer_keep_reads): one chunk per call, blocks of the pool
    struct KeptReads { uint8_t* store = nullptr; long long* start = nullptr; int* length = nullptr; int n = 0; };
    std::vector<KeptReads> kept;
    std::string err;
};

struct hlala_batch {
    hlala_ctx* ctx = nullptr;
    DevBatch B{};
    DevBatch* dB = nullptr;       // device copy of B
    std::vector<void*> allocs;
    int staged = 0;   // bit0 seeds available, bit1 extended, bit2 paired
    bool prepared = false;           // the filters and the position order ran when the batch was created; B.n_rows = chains that hold column rows (batch.h: chain_row)
    int n_rows_host = 0;             // ... read back with the upload's synchronisation
    bool outputs_ready = false;      // the output arrays (50 GB for a 1 M-pair batch) exist: allocated by the first stage call, not by hlala_batch_create (ensure_outputs)
    bool side_used = false;      // the last extend of this batch ran its wide classes on the side stream (their times are between the evSide events)
    bool side_pending = false;   // ... and hlala_pair_chains has yet to enqueue the second pairing pass behind them
    bool tail_pooled = false;    // the batch waits in its context's tail pool: its deferred pairs are complete after flush_tail (readers and stage calls flush first)
    bool side_inflight = false;  // work of this batch may still be running on the side stream: evDone orders everything that touches the batch after it
    hipEvent_t evDone = nullptr;
    hipEvent_t evMain = nullptr; bool mainValid = false;      // end of the last work of this batch on the main stream (readers on `rs` wait for it)
    // timing events of THIS batch (created with its first stage call): ev = start / end per stage, [7] / [6] / [10] / [8] = before the 16-lane class / after it /
    // after the 64-lane class / after the last class; evC = start / end of each DP class on the stream it ran on; evSide[0] fork point on the main stream,
    // [1] first side-stream class starts, [6] second pairing pass done
    hipEvent_t ev[14]{}; hipEvent_t evC[7][2]{}; hipEvent_t evSide[8]{}; hipEvent_t evJF = nullptr; /* end of the jump-free instantiation of the 16-lane class */ hipEvent_t evBand[2]{}; /* the band kernel */ bool band_used = false; hipEvent_t evBand2[2]{}; bool band2_used = false; bool eventsMade = false;
    uint32_t first_chain = 0;    // absolute index of the batch's chain 0 in the caller's numbering (hlala_batch_set_first_chain): offsets the random seeds
    float ms[3] = {0, 0, 0};
};

// Every entry point runs with the context's device current and restores the caller's device on return: several contexts (one per GPU)
// may live in one process, and the host program (or torch) may switch devices between calls.
struct DevGuard {
    int prev = -1, want = -1;
    explicit DevGuard(int device) : want(device) { if(device >= 0 && hipGetDevice(&prev) == hipSuccess && prev != device) (void)hipSetDevice(device); else prev = -1; }
    ~DevGuard() { if(prev >= 0 && prev != want) (void)hipSetDevice(prev); }
    DevGuard(const DevGuard&) = delete; DevGuard& operator=(const DevGuard&) = delete;
};
#define DEV_GUARD(c) DevGuard dev_guard_((c) ? (c)->device : -1)

#define HIP_TRY(ctx, call) do { hipError_t e_ = (call); if(e_ != hipSuccess) { (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e_); return HLALA_E_DEVICE; } } while(0)
// the same inside a function that owns temporaries or a half-built object: `cleanup` (a lambda int -> int) releases them and passes the code through
#define HIP_TRY_F(ctx, call, cleanup) do { hipError_t e_ = (call); if(e_ != hipSuccess) { (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e_); return cleanup(HLALA_E_DEVICE); } } while(0)

// everything on the main stream that reads or rewrites a batch goes behind the side-stream work of its last fused alignment
static int flush_tail(hlala_ctx* c);
static int join_side(hlala_ctx* c, hlala_batch* b)
{
    if(b->tail_pooled) { int rf = flush_tail(c); if(rf) return rf; }
    if(b->side_inflight) {
        HIP_TRY(c, hipStreamWaitEvent(c->stream, b->evDone, 0));
        // evMain is what hlala_batch_destroy and the readers wait for once side_inflight is cleared: it has to lie BEHIND the wait just queued even if the
        // stage call that joined returns early (state / launch error) and never reaches its own mark_main
        if(b->evMain) { HIP_TRY(c, hipEventRecord(b->evMain, c->stream)); b->mainValid = true; b->side_inflight = false; }
    }
    return HLALA_OK;
}
static int batch_events(hlala_ctx* c, hlala_batch* b)
{
    if(b->eventsMade) return HLALA_OK;
    for(int i = 0; i < 14; i++) HIP_TRY(c, hipEventCreate(&b->ev[i]));
    for(int i = 0; i < 8; i++) HIP_TRY(c, hipEventCreate(&b->evSide[i]));
    for(int i = 0; i < 14; i++) HIP_TRY(c, hipEventCreate(&b->evC[i / 2][i % 2]));
    HIP_TRY(c, hipEventCreate(&b->evJF));
    for(int i = 0; i < 2; i++) HIP_TRY(c, hipEventCreate(&b->evBand[i]));
    for(int i = 0; i < 2; i++) HIP_TRY(c, hipEventCreate(&b->evBand2[i]));
    HIP_TRY(c, hipEventCreateWithFlags(&b->evMain, hipEventDisableTiming));
    b->eventsMade = true;
    return HLALA_OK;
}
static int mark_main(hlala_ctx* c, hlala_batch* b)       // end of a stage call: what readers of the batch on the reader stream wait for
{
    HIP_TRY(c, hipEventRecord(b->evMain, c->stream)); b->mainValid = true;
    return HLALA_OK;
}
// A call that only reads a batch (or works on caller data) runs on the context's reader stream, behind the batch's own work on the main and the side
// stream -- not behind whatever the caller queued for other batches since.  (Calls on one context are serialised by the caller: `active` is plain state.)
struct ReaderScope {
    hlala_ctx* c; int rc = HLALA_OK;
    ReaderScope(hlala_ctx* c_, hlala_batch* b) : c(c_)
    {
        if(!c) return;
        if(b && b->tail_pooled) { rc = flush_tail(c); if(rc) return; }          // the batch's tail classes are still pooled: run them now (with whatever the pool holds)
        c->active = c->rs;
        hipError_t e = hipSuccess;
        if(b && b->mainValid) e = hipStreamWaitEvent(c->rs, b->evMain, 0);
        if(e == hipSuccess && b && b->side_inflight) e = hipStreamWaitEvent(c->rs, b->evDone, 0);
        if(e != hipSuccess) { c->err = std::string("hipStreamWaitEvent: ") + hipGetErrorString(e); rc = HLALA_E_DEVICE; }
    }
    ~ReaderScope() { if(c) c->active = c->stream; }
    ReaderScope(const ReaderScope&) = delete; ReaderScope& operator=(const ReaderScope&) = delete;
};
struct UploadScope {
    hlala_ctx* c;
    explicit UploadScope(hlala_ctx* c_) : c(c_) { if(c) c->active = c->up; }
    ~UploadScope() { if(c) c->active = c->stream; }
    UploadScope(const UploadScope&) = delete; UploadScope& operator=(const UploadScope&) = delete;
};

// hipMalloc, or a block of a destroyed batch that is large enough and wastes at most a quarter
// Sizes are rounded up to classes a sixteenth of a power of two apart (at most 6 % more than asked for): the arrays of consecutive batches differ by a fraction of a
// per cent in size, and with exact sizes a batch kept finding the blocks of the batch before it a little too small for some of its arrays -- a few hipMalloc and
// hipFree of gigabyte blocks in every step, 8-16 ms per hlala_align_batch in the first process on a freshly booted device (0.4 ms once the driver has handed the
// memory out before), which is the step the next batch's kernels are launched from.
static size_t pool_class(size_t bytes)
{
    bytes = (bytes + 255) & ~(size_t)255;
    if(bytes <= 4096) return bytes;
    const int lg = 63 - __builtin_clzll((unsigned long long)bytes);
    const size_t g = (size_t)1 << (lg - 4);
    return (bytes + g - 1) & ~(g - 1);
}
// Every live context is registered: a context parks up to 176 GB of a 288 GB device, and an allocation of ANOTHER context on the same device (a second hlala_create,
// its slabs, its batches) must be able to get that memory back.  A context's pool is touched by its own thread (pool_malloc / pool_release) and, on an out-of-memory
// error anywhere on the device, by the thread that met the error: pool_mu orders the two.
static std::mutex g_ctx_mu;
static std::vector<hlala_ctx*> g_ctxs;
static void pool_trim_locked(hlala_ctx* c)
{
    for(auto& kv : c->pool) { c->block_bytes.erase(kv.second); (void)hipFree(kv.second); }
    c->pool.clear(); c->pool_bytes = 0;
}
// hipMalloc; on failure the blocks parked by every context of this device are released and the call is tried once more (`self`: the caller's context when it
// already holds its own pool_mu, else null)
static hipError_t device_malloc_retry(int device, hlala_ctx* self, void** p, size_t bytes)
{
    hipError_t e = hipMalloc(p, bytes);
    if(e == hipSuccess) return e;
    (void)hipGetLastError();
    bool freed = false;
    {
        std::lock_guard<std::mutex> g(g_ctx_mu);
        for(hlala_ctx* o : g_ctxs) {
            if(o->device != device) continue;
            if(o == self) { if(!o->pool.empty()) { pool_trim_locked(o); freed = true; } continue; }
            std::lock_guard<std::mutex> g2(o->pool_mu);
            if(!o->pool.empty()) { pool_trim_locked(o); freed = true; }
        }
    }
    if(!freed) return e;
    return hipMalloc(p, bytes);
}
static int pool_malloc(hlala_ctx* c, void** out, size_t bytes)
{
    bytes = pool_class(bytes);
    std::lock_guard<std::mutex> g(c->pool_mu);
    auto it = c->pool.lower_bound(bytes);
    if(it != c->pool.end() && it->first <= bytes + bytes / 4 + 4096) {
        *out = it->second; c->pool_bytes -= it->first; c->pool.erase(it);
        return 0;
    }
    void* p = nullptr;
    hipError_t e = device_malloc_retry(c->device, c, &p, bytes);       // out of memory with blocks parked (here or in another context of the device): released, then retried
    if(e != hipSuccess) { c->err = std::string("hipMalloc: ") + hipGetErrorString(e); return HLALA_E_DEVICE; }
    c->block_bytes[p] = bytes;
    *out = p;
    return 0;
}
static void pool_release(hlala_ctx* c, void* p)
{
    if(!p) return;
    std::lock_guard<std::mutex> g(c->pool_mu);
    auto it = c->block_bytes.find(p);
    if(it == c->block_bytes.end()) { (void)hipFree(p); return; }
    if(c->pool_bytes + it->second > c->pool_cap) { c->block_bytes.erase(it); (void)hipFree(p); return; }      // keep at most 176 GB parked (three 1 M-pair batches' arrays: a caller with three sets of outputs live gives them all back between two runs)
    c->pool.emplace(it->second, p); c->pool_bytes += it->second;
}

constexpr size_t PIN_GRANULE = (size_t)64 << 20;
template <class T>
static int dev_upload(hlala_ctx* c, std::vector<void*>& allocs, const T* host, size_t n, T** out)
{
    *out = nullptr;
    size_t bytes = (n ? n : 1) * sizeof(T);
    void* p = nullptr;
    { int rc_ = pool_malloc(c, &p, bytes); if(rc_) return rc_; }
    allocs.push_back(p);
    // A caller's array may be page-locked piece by piece (hlala_seed_batch_pin(S, 2): pieces that begin and end on multiples of PIN_GRANULE bytes of address); the
    // runtime refuses a copy whose source straddles two registrations, so no copy crosses such an address -- a dozen copies per GB instead of one.
    if(n && host) {
        const char* src = (const char*)host; char* dst = (char*)p; size_t left = n * sizeof(T);
        while(left) {
            const size_t toEdge = PIN_GRANULE - (size_t)((uintptr_t)src & (PIN_GRANULE - 1)), k = left < toEdge ? left : toEdge;
            HIP_TRY(c, hipMemcpyAsync(dst, src, k, hipMemcpyHostToDevice, c->active));
            src += k; dst += k; left -= k;
        }
    }
    *out = (T*)p;
    return 0;
}
template <class T>
static int dev_alloc(hlala_ctx* c, std::vector<void*>& allocs, size_t n, T** out, bool zero = false)
{
    *out = nullptr;
    size_t bytes = (n ? n : 1) * sizeof(T);
    void* p = nullptr;
    { int rc_ = pool_malloc(c, &p, bytes); if(rc_) return rc_; }
    allocs.push_back(p);
    if(zero) HIP_TRY(c, hipMemsetAsync(p, 0, bytes, c->active));
    *out = (T*)p;
    return 0;
}
#define UP(vec, field) do { int rc_ = dev_upload(c, c->allocs, (vec).data(), (vec).size(), &tmp_##field); if(rc_) return rc_; } while(0)

// ---- host-side constant tables (host libm => bit-identical to a CPU evaluation of the reference formulas)

/* Utilities::PhredToPCorrect, Utilities.cpp:357-377 */
static double host_PhredToPCorrect(unsigned char q)
{
    if(q == 0) return -1;
    int illuminaPhred = (int)q - 33;
    double log10_pWrong = (double)illuminaPhred / (double)-10;
    double pWrong = exp(log(10) * log10_pWrong);
    return 1 - pWrong;
}
/* Utilities::PCorrectToPhred (Utilities.cpp:178-203) as a function of pWrong */
static int host_phred_of_pwrong(double pWrong)
{
    if(pWrong == 0) pWrong = 1e-100;
    double phred1 = -10.0 * log10(pWrong);
    if((phred1 + 33) > 255) phred1 = 255 - 33;
    return (int)round(phred1 + 33);
}
/* boost::math::pdf(normal) closed form, see oracle header note; processBAM.cpp:2343, 3446 */
static double host_normal_pdf(double mean, double sd, double x)
{
    double exponent = x - mean;
    exponent *= -exponent;
    exponent /= 2 * sd * sd;
    double result = exp(exponent);
    result /= sd * sqrt(2 * 3.141592653589793238462643383279502884);
    return result;
}

static int build_tables(hlala_ctx* c)
{
    DevTables T;
    memset(&T, 0, sizeof(T));
    for(int q = 0; q < 256; q++) {
        double pCorrect = host_PhredToPCorrect((unsigned char)q);
        if(q < 33) pCorrect = host_PhredToPCorrect(33);         // the reference asserts illuminaPhred >= 0; never indexed for valid input
        if(pCorrect > 0.999) pCorrect = 0.999;                    // conservativeReadQualities, extensionAligner.cpp:128-131
        if(pCorrect == 0) pCorrect = 0.00001;
        T.ll_match[q] = log(pCorrect);
        double pIncorrect = 1 - pCorrect; pIncorrect *= (1.0 / 3.0);
        T.ll_mismatch[q] = log(pIncorrect);
        T.pcorrect[q] = host_PhredToPCorrect((unsigned char)q);
    }
    double rate = c->params.long_read_mode ? log(0.075) : log(0.001);
    T.rate_indel = rate;
    T.rate_ins_quarter = rate + log(1.0 / 4.0);
    T.rate_match_mismatch = log(1 - exp(rate) - exp(rate));
    // PCorrectToPhred thresholds: phred_thr[k] = largest pWrong (as a double) that still maps to a Phred char >= k
    for(int k = 0; k < 256; k++) {
        if(host_phred_of_pwrong(1.0) >= k) { T.phred_thr[k] = 1.0; continue; }
        double lo = 1e-300, hi = 1.0;      // f(lo) >= k (= 255), f(hi) < k
        uint64_t blo, bhi; memcpy(&blo, &lo, 8); memcpy(&bhi, &hi, 8);
        while(bhi - blo > 1) {
            uint64_t mid = blo + (bhi - blo) / 2; double m; memcpy(&m, &mid, 8);
            if(host_phred_of_pwrong(m) >= k) blo = mid; else bhi = mid;
        }
        double r; memcpy(&r, &blo, 8);
        T.phred_thr[k] = r;
    }
    // insert-size log pdf over every integer distance with a positive density
    double mean = c->params.insert_mean, sd = c->params.insert_sd;
    std::vector<double> tbl;
    T.is_dmin = 0; T.is_n = 0;
    if(sd > 0) {
        T.is_penalty = log(host_normal_pdf(mean, sd, mean + 8 * sd));
        long long dmin = (long long)floor(mean - 40 * sd) - 2, dmax = (long long)ceil(mean + 40 * sd) + 2;
        if(dmax - dmin > 50000000) { c->err = "insert size sd too large for the log-pdf table"; return HLALA_E_ARG; }
        if(host_normal_pdf(mean, sd, (double)dmin) > 0 || host_normal_pdf(mean, sd, (double)dmax) > 0) { c->err = "log-pdf table does not reach zero density"; return HLALA_E_ARG; }
        tbl.resize((size_t)(dmax - dmin + 1));
        for(long long d = dmin; d <= dmax; d++) {
            double p = host_normal_pdf(mean, sd, (double)d);
            tbl[(size_t)(d - dmin)] = (p <= 0) ? T.is_penalty : log(p);             // processBAM.cpp:3447-3464
        }
        T.is_dmin = (int)dmin; T.is_n = (int)tbl.size();
    }
    int rc = dev_upload(c, c->allocs, tbl.data(), tbl.size(), &c->d_islog); if(rc) return rc;
    T.is_logpdf = c->d_islog;
    rc = dev_upload(c, c->allocs, &T, 1, &c->dT); if(rc) return rc;
    return 0;
}

template <class T>
static int dl(hlala_ctx* c, T* host, const T* dev, size_t n)
{
    if(!host || !n) return 0;
    HIP_TRY(c, hipMemcpyAsync(host, dev, n * sizeof(T), hipMemcpyDeviceToHost, c->active));
    return 0;
}

extern "C" {

const char* hlala_last_error(const hlala_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

int hlala_create(hlala_ctx** out, int device, void* stream, const hlala_graph_desc* graph, const hlala_contigs_desc* contigs, const hlala_params* params)
{
    if(!out || !graph || !params) { g_create_error = "null argument"; return HLALA_E_ARG; }
    *out = nullptr;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if(e != hipSuccess || ndev <= 0 || device >= ndev) {
        g_create_error = std::string("no HIP device available (") + (e != hipSuccess ? hipGetErrorString(e) : "device index out of range") + "); this library has no CPU fallback";
        return HLALA_E_DEVICE;
    }
    hlala_ctx* c = new hlala_ctx();
    c->device = device; c->stream = (hipStream_t)stream; c->active = c->stream; c->params = *params;
    auto fail = [&](int rc) { g_create_error = c->err; hlala_destroy(c); return rc; };
    if(c->params.max_columns < 16 || c->params.max_columns > 65536) { c->err = "params.max_columns out of range"; return fail(HLALA_E_ARG); }
    DevGuard dev_guard_(device);       // the caller's current device is restored on return
    { int cur_ = -1; if(hipGetDevice(&cur_) != hipSuccess || cur_ != device) { c->err = "hipSetDevice failed"; return fail(HLALA_E_DEVICE); } }
    const bool keepUnitJumps = [] { const char* e = getenv("HLALA_UNIT_JUMPS"); return e && atoi(e) != 0; }();       // (A/B and parity: the one-edge gap paths stay in the device's jump tables)
    std::string ferr = flatten_graph(graph, contigs, c->F, keepUnitJumps);
    if(!ferr.empty()) { c->err = ferr; return fail(HLALA_E_GRAPH); }
    FlatGraph& F = c->F;
    if(F.N >= (1 << 28) || F.L >= (1 << 24)) { c->err = "graph exceeds 2^28 nodes or 2^24 levels (DP cell key layout)"; return fail(HLALA_E_CAPACITY); }
    // the push index of a DP candidate holds the rank of its edge among the PARALLEL edges (same two nodes) in 7 bits: a node may have any number of
    // edges and gap-path jumps, but more than 127 parallel ones between two nodes are refused here instead of dropping pairs later
    if(F.max_parallel > DP_MAX_PARALLEL) {
        c->err = "graph has " + std::to_string(F.max_parallel) + " parallel edges or gap paths between one pair of nodes: more than the " + std::to_string(DP_MAX_PARALLEL) + " the DP's push-index field ranks";
        return fail(HLALA_E_CAPACITY);
    }
    if(F.max_nodes_per_level > PROJ_NODES) { c->err = "more nodes in one level than this build holds in LDS (PROJ_NODES)"; return fail(HLALA_E_CAPACITY); }
    if(c->params.max_columns > PROJL_CAP) { c->err = "params.max_columns exceeds the column capacity of this build (16384)"; return fail(HLALA_E_ARG); }
    DevGraph& G = c->G;
    G.L = F.L; G.N = F.N; G.E = F.E; G.P = (int)F.path_len.size();
    std::vector<uint8_t> edge_label(graph->edge_label, graph->edge_label + graph->n_edges);
    int rc = 0;
#define UPG(field, vec) do { rc = dev_upload(c, c->allocs, (vec).data(), (vec).size(), (std::remove_const<std::remove_pointer<decltype(G.field)>::type>::type**)&G.field); if(rc) return fail(rc); } while(0)
    UPG(level_off, F.level_off); UPG(node_level, F.node_level); UPG(node_orig, F.node_orig);
    UPG(out_off, F.out_off); UPG(out_to, F.out_to); UPG(out_label, F.out_label); UPG(out_eid, F.out_eid);
    UPG(in_off, F.in_off); UPG(in_from, F.in_from); UPG(in_label, F.in_label); UPG(in_eid, F.in_eid); UPG(in_rec, F.in_rec); UPG(level_fast, F.level_fast);
    UPG(edge_from_new, F.edge_from_new); UPG(edge_to_new, F.edge_to_new); UPG(edge_label, edge_label);
    // (the device's jump tables hold the gap paths of two and more edges only: a one-edge path is a no-op for the DP, flat_graph.hpp)
    UPG(jf_off, F.djf_off); UPG(jf_node, F.djf_node); UPG(jf_path, F.djf_path);
    UPG(jb_off, F.djb_off); UPG(jb_node, F.djb_node); UPG(jb_path, F.djb_path);
    UPG(jf_lvl, F.djf_lvl); UPG(jb_lvl, F.djb_lvl);
    UPG(jfree_out, F.jfree_out); UPG(jfree_in, F.jfree_in);
    UPG(lin_label, F.lin_label); UPG(lin_out, F.lin_out); UPG(lin_in, F.lin_in); UPG(lin_eid, F.lin_eid);
    UPG(trk_w_out, F.trk_w_out); UPG(trk_w_in, F.trk_w_in); UPG(trk_out, F.trk_out); UPG(trk_in, F.trk_in); UPG(trk_j_out, F.trk_j_out); UPG(trk_j_in, F.trk_j_in); UPG(trk_jp_out, F.trk_jp_out); UPG(trk_jp_in, F.trk_jp_in);
    UPG(out_prank, F.out_prank); UPG(in_prank, F.in_prank); UPG(jf_prank, F.jf_prank); UPG(jb_prank, F.jb_prank);
    { int* p_ = nullptr; rc = dev_upload(c, c->allocs, F.nrec_out.data(), F.nrec_out.size(), &p_); if(rc) return fail(rc); G.nrec_out = (const int4*)p_;
      rc = dev_upload(c, c->allocs, F.nrec_in.data(), F.nrec_in.size(), &p_); if(rc) return fail(rc); G.nrec_in = (const int4*)p_; }
    UPG(path_len, F.path_len); UPG(path_edges, F.path_edges);
    {
        std::vector<long long> po(F.path_off.begin(), F.path_off.end());
        rc = dev_upload(c, c->allocs, po.data(), po.size(), (long long**)&G.path_off); if(rc) return fail(rc);
        std::vector<long long> lo(F.lp_off.begin(), F.lp_off.end());
        rc = dev_upload(c, c->allocs, lo.data(), lo.size(), (long long**)&G.lp_off); if(rc) return fail(rc);
    }
    UPG(gap_stretch, F.gap_stretch); UPG(lp_seqid, F.lp_seqid); UPG(lp_pos, F.lp_pos);
#undef UPG
    if(contigs && contigs->n_contigs > 0) {
        c->n_contigs = contigs->n_contigs;
        c->contig_off.assign(contigs->contig_off, contigs->contig_off + contigs->n_contigs + 1);
        size_t total = (size_t)c->contig_off.back();
        std::vector<long long> co(c->contig_off.begin(), c->contig_off.end());
        rc = dev_upload(c, c->allocs, co.data(), co.size(), &c->d_contig_off); if(rc) return fail(rc);
        rc = dev_upload(c, c->allocs, contigs->contig_seq, total, &c->d_contig_seq); if(rc) return fail(rc);
        rc = dev_upload(c, c->allocs, contigs->contig_level, total, &c->d_contig_level); if(rc) return fail(rc);
    }
    rc = build_tables(c); if(rc) return fail(rc);
    rc = dev_upload(c, c->allocs, &c->G, 1, &c->dG); if(rc) return fail(rc);
    // scratch slabs: one per resident wavefront (persistent grid, dynamic work distribution)
    hipDeviceProp_t prop;
    if(hipGetDeviceProperties(&prop, device) != hipSuccess) { c->err = "hipGetDeviceProperties failed"; return fail(HLALA_E_DEVICE); }
    int cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    // DP slab pools, one per class and zeroed once: a slab's early-cell table is cleared by the first DP call that needs it and kept clean from then on
    // (kernel_dp.hip: DP_SLAB_READY), which only holds while no other class lays its own arrays over the same memory
    auto slab_pool = [&](char** out, size_t bytes, const char* what) -> int {
        if(device_malloc_retry(c->device, nullptr, (void**)out, bytes) != hipSuccess) { c->err = std::string("hipMalloc(") + what + ") failed"; return HLALA_E_DEVICE; }
        c->allocs.push_back(*out);
        if(hipMemsetAsync(*out, 0, bytes, c->active) != hipSuccess) { c->err = std::string("hipMemset(") + what + ") failed"; return HLALA_E_DEVICE; }
        return 0;
    };
    c->tiny_grid = cus * 4 * DpTiny::WAVES;
    if(const char* e = getenv("HLALA_TINY_WAVES_PER_CU")) { const int w = atoi(e); if(w >= 1 && w <= 4 * DpTiny::WAVES) c->tiny_grid = cus * w; }      // (experiment: blocks of the 16-lane kernel per CU, tools/gpu_tiny_waves.sh)
    c->tiny_slab_bytes = dp_slab_bytes<DpTiny>();
    c->jf_grid = cus * 4 * DpTinyJF::WAVES;
    c->band_grid = cus * 20;          // a few KB of LDS per block, five waves per SIMD (96 VGPRs, nothing spilled)
    if(const char* e = getenv("HLALA_DP_BAND")) { if(atoi(e) == 0) c->band_grid = 0; }      // (A/B and parity: every call in the hashed-frontier classes)
    if(const char* e = getenv("HLALA_DP_BAND_RISKY")) c->band_risky = atoi(e) != 0;
    if(const char* e = getenv("HLALA_ROWS_ALL")) c->rows_all = atoi(e) != 0;
    if(const char* e = getenv("HLALA_SIDE_AFTER_PAIR")) c->side_after_pair = atoi(e) != 0;
    if(const char* e = getenv("HLALA_POOL_CAP_GB")) { const long g = atol(e); if(g >= 0 && g <= 1024) c->pool_cap = (size_t)g << 30; }
    if(const char* e = getenv("HLALA_TAIL_POOL")) { const int k = atoi(e); if(k >= 1 && k <= DP_POOL_MAX) c->tail_pool_k = k; }      // (experiments and the parity suite: hlala_set_tail_pool without touching the caller)
    if(const char* e = getenv("HLALA_DP_BAND_MARGIN")) { const int m = atoi(e); if(m >= 0 && m <= 24) c->band_margin = m; }
    // The two-track band kernels are bit-exact and SLOWER than the hashed-frontier classes they would relieve (profiles/r06_experiments.txt 6: 1 000 vector instructions
    // per iteration for two bands of two tracks -- the per-call cost of the hashed machine): not part of the default path.  HLALA_DP_BAND2=1 switches them on (the parity
    // suite runs them: tests/test_gpu_align.py); the slabs are only allocated then.
    c->band2_grid = 0;
    if(const char* e = getenv("HLALA_DP_BAND2")) { if(atoi(e) != 0) c->band2_grid = cus * 6; }      // 19-25 KB of LDS per block (the ring of the early band's cells)
    if(const char* e = getenv("HLALA_DP_BAND2_MAXJ")) { const int m = atoi(e); if(m >= 1 && m <= B2_MAXJ64) c->band2_maxj = m; }
    if(const char* e = getenv("HLALA_DP_BAND2_MARGIN")) { const int m = atoi(e); if(m >= 0 && m <= 200) c->band2_margin = m; }
    if(const char* e = getenv("HLALA_DP_BAND2_WAVES")) { const int w = atoi(e); if(w >= 1 && w <= 16 && c->band2_grid) c->band2_grid = cus * w; }
    if(c->band2_grid) { char* p_ = nullptr; if((rc = slab_pool(&p_, (size_t)c->band2_grid * (size_t)B2_MAXD * 64 * sizeof(u64), "two-track band slabs"))) return fail(rc); c->band2_slabs = (u64*)p_; }
    if(const char* e = getenv("HLALA_DP_BAND_WAVES")) { const int w = atoi(e); if(w >= 1 && w <= 32 && c->band_grid) c->band_grid = cus * w; }
    if(const char* e = getenv("HLALA_DP_JF")) { if(atoi(e) == 0) c->jf_grid = 0; }      // (A/B: every call in the general instantiation -- the kernels' lists are built either way)
    if(const char* e = getenv("HLALA_DP_JF_MARGIN")) { const int m = atoi(e); if(m >= 0 && m <= 200) c->jf_margin = m; }      // (A/B: levels beyond the read bases left that a jump-free call may reach)
    c->ext_grid = cus * 20;
    c->mid_grid = cus * 16; c->mid_slab_bytes = dp_slab_bytes<DpMid>();
    c->retry_grid = cus;
    c->broad_grid = cus * 3;         // three DpBroad blocks per CU (48 KB of LDS each), slabs of the large layout

    c->wide_grid = cus * 7;          // LDS: seven DpWide blocks per CU (22 KB each); slabs of the 64-lane layout
    if(const char* e = getenv("HLALA_DP_WIDE_BLOCKS")) { const int w = atoi(e); if(w >= 1 && w <= 7) c->wide_grid = cus * w; }      // (experiment: the class runs on the side stream since round 5 -- how much LDS it may hold beside the main stream's kernels)
    c->stitch_grid = cus * 20;        // k_stitch_chains: five waves per SIMD (92 VGPRs, nothing spilled)
    if(const char* e = getenv("HLALA_STITCH_WAVES")) { const int w = atoi(e); if(w >= 1 && w <= 20) c->stitch_grid = cus * w; }      // (experiments: waves per CU and chains per wave and round of k_stitch_chains)
    if(const char* e = getenv("HLALA_STITCH_BY_ROW")) c->stitch_by_row = atoi(e) != 0;
    if(const char* e = getenv("HLALA_STITCH_DRAW")) { const int d = atoi(e); if(d >= 1 && d <= 64) c->stitch_draw = d; }
    c->ext_slab_bytes = dp_slab_bytes<DpSmall>() > dp_slab_bytes<DpWide>() ? dp_slab_bytes<DpSmall>() : dp_slab_bytes<DpWide>();
    c->large_slab_bytes = dp_slab_bytes<DpLarge>() > dp_slab_bytes<DpBroad>() ? dp_slab_bytes<DpLarge>() : dp_slab_bytes<DpBroad>();       // one / three blocks per CU: a few MB each
    c->huge_grid = cus / 4 > 0 ? cus / 4 : 1;      // the in-memory backstop class: a handful of DP calls per million pairs
    if(const char* e = getenv("HLALA_DP_HUGE_BLOCKS")) { const int v = atoi(e); if(v >= 1 && v <= 2 * cus) c->huge_grid = v; }      // (experiment: blocks of the in-memory class -- 128 calls per million pairs on 64 blocks run in two rounds)
    c->huge_slab_bytes = dp_inmemory_bytes<DpHuge>();
    if((rc = slab_pool(&c->tiny_slabs, c->tiny_slab_bytes * (size_t)(64 / DpTiny::GW) * (size_t)(c->tiny_grid > c->jf_grid ? c->tiny_grid : c->jf_grid), "16-lane DP slabs"))) return fail(rc);
    if((rc = slab_pool(&c->mid_slabs, c->mid_slab_bytes * (size_t)(64 / DpMid::GW) * (size_t)c->mid_grid, "32-lane DP slabs"))) return fail(rc);
    if((rc = slab_pool(&c->ext_slabs, c->ext_slab_bytes * (size_t)c->ext_grid, "64-lane DP slabs"))) return fail(rc);
    if((rc = slab_pool(&c->wide_slabs, c->ext_slab_bytes * (size_t)c->wide_grid, "wide-class DP slabs"))) return fail(rc);
    if((rc = slab_pool(&c->large_slabs, c->large_slab_bytes * (size_t)(c->broad_grid + c->retry_grid), "large-class DP slabs"))) return fail(rc);       // broad blocks first, then the large ones
    if((rc = slab_pool(&c->huge_slabs, c->huge_slab_bytes * (size_t)c->huge_grid, "in-memory DP class"))) return fail(rc);
    // k_project_chains<384 columns>: 143 VGPRs = three waves per SIMD = 12 resident blocks per CU (its 11.7 KB of LDS would allow 13); the 512-column
    // layout: 173 VGPRs = two per SIMD (-Rpass-analysis=kernel-resource-usage; a cap of 128 VGPRs for a fourth wave costs 287 spilled SGPRs and wins one block)
    c->proj_grid = cus * (c->params.max_columns <= PROJ_CAP_SHORT ? 12 : 8); c->pair_grid = cus * 16; c->pair_lean_grid = cus * 24;      // k_pair_multi<., false>: four waves per SIMD (125 registers, nothing spilled); k_pair_chains (0.7 KB of LDS, 80 registers): six
    if(const char* e = getenv("HLALA_PAIR_WAVES")) { const int w = atoi(e); if(w >= 1 && w <= 32) c->pair_grid = cus * w; }      // (experiments)
    if(const char* e = getenv("HLALA_PROJ_WAVES")) { const int w = atoi(e); if(w >= 1 && w <= 14) c->proj_grid = cus * w; }      // (experiment: waves of the projection kernel per CU)
    { int rcp = dev_alloc(c, c->allocs, (size_t)2 * c->pair_grid * PAIR_COMB, &c->pair_scratch, false); if(rcp) return fail(rcp); }
    if(c->params.max_columns > PROJ_CAP) {       // long reads: the projection keeps its column / window arrays in HBM, fewer and bigger blocks
        c->proj_grid = cus * 20;       // (round 6: 20 waves per CU -- 95 VGPRs, five per SIMD, 6.3 KB of LDS each; round 5: 16 waves per CU -- 103 VGPRs, four per SIMD; with 4 per CU the kernel ran one wave per SIMD, waiting 72 % of its cycles: 71 k -> 82 k reads/s in batches of 10 000, 223 k in one batch of 50 000)
        if(const char* e = getenv("HLALA_PROJ_LONG_WAVES")) { const int w = atoi(e); if(w >= 1 && w <= 20) c->proj_grid = cus * w; }      // (experiment: waves of the long-read projection per CU)
        c->proj_long_slab_bytes = proj_long_slab_bytes();
        if(!getenv("HLALA_STITCH_DRAW")) c->stitch_draw = 1;       // rows of 16 384 columns: one chain per draw (12: 11.0 ms per 50 000 reads, 1: 8.8; a batch of 12 500: 7.9 -> 4.0 ms)
        if(device_malloc_retry(c->device, nullptr, (void**)&c->proj_long_slabs, c->proj_long_slab_bytes * (size_t)c->proj_grid) != hipSuccess) { c->err = "hipMalloc(long-read projection slabs) failed"; return fail(HLALA_E_DEVICE); }
        c->allocs.push_back(c->proj_long_slabs);
    }
    c->proj_slab_bytes = proj_slab_bytes_host(c->params.max_columns, F.max_nodes_per_level);
    if(device_malloc_retry(c->device, nullptr, (void**)&c->proj_slabs, c->proj_slab_bytes * (size_t)c->proj_grid) != hipSuccess) { c->err = "hipMalloc(projection slabs) failed"; return fail(HLALA_E_DEVICE); }
    c->allocs.push_back(c->proj_slabs);
    if(!c->proj_long_slabs) {
        const char* e = getenv("HLALA_RETHREAD");
        if(!(e && atoi(e) == 0)) {
            // a chain of that kernel has at most RT_SN nodes per level: its window holds at most max_columns * RT_SN nodes
            size_t ent = (size_t)c->params.max_columns * (size_t)RT_SN; if(ent < 1024) ent = 1024;
            c->rethread_slab_bytes = (ent * sizeof(ChoiceRec) + 255) & ~(size_t)255;
            if(c->rethread_slab_bytes > c->proj_slab_bytes) c->rethread_slab_bytes = c->proj_slab_bytes;
            int wv = 24; if(const char* w = getenv("HLALA_RETHREAD_WAVES")) { const int v = atoi(w); if(v >= 1 && v <= 24) wv = v; }
            c->rethread_grid = cus * wv;
            if(device_malloc_retry(c->device, nullptr, (void**)&c->rethread_slabs, c->rethread_slab_bytes * (size_t)c->rethread_grid) != hipSuccess) { c->err = "hipMalloc(re-threading slabs) failed"; return fail(HLALA_E_DEVICE); }
            c->allocs.push_back(c->rethread_slabs);
        }
    }
    // position buckets: a few hundred levels each, at most 16 384 of them (the scan is one block)
    { const char* e = getenv("HLALA_LOCALITY");
      if(!(e && atoi(e) == 0)) { int sh = 8; if(e && atoi(e) >= 2 && atoi(e) <= 20) sh = atoi(e); while((F.L >> sh) + 2 > 16384) sh++; c->order_shift = sh; c->order_nb = (F.L >> sh) + 2; } }
    if(const char* e = getenv("HLALA_PROJ_LONG_STAGGER")) c->proj_long_stagger = atoi(e);
    if(const char* e = getenv("HLALA_LONG_CHUNK_NODES")) { const int v = atoi(e); if(v >= 1) c->long_chunk_nodes = v; }
    if(const char* e = getenv("HLALA_LONG_MAXSEGS")) { const int v = atoi(e); if(v >= 0) c->long_max_segs = v; }
    if(c->proj_long_slabs && c->order_nb > 0) { const char* e = getenv("HLALA_LONG_ORDER"); c->order_cost = (e && atoi(e) == 0) ? 0 : 1; }
    // (the debug buffer is DEVICE memory since round 6: host-mapped, every counter a kernel added to it was an atomic over PCIe -- the wavefronts of the long-read projection queued
    //  behind each other's, and the clocks they were meant to read showed that queue)
    if(getenv("HLALA_DEBUG")) { if(hipMalloc((void**)&c->dbg_host, 8192 * sizeof(int)) != hipSuccess) c->dbg_host = nullptr; else { (void)hipMemset(c->dbg_host, 0, 8192 * sizeof(int)); c->allocs.push_back(c->dbg_host); } }
    if(hipEventCreateWithFlags(&c->evSideTail, hipEventDisableTiming) != hipSuccess) { c->err = "hipEventCreate failed"; return fail(HLALA_E_DEVICE); }
    { int prLow = 0, prHigh = 0; (void)hipDeviceGetStreamPriorityRange(&prLow, &prHigh);       // (numerically greatest = lowest priority)
      // the upload and the reader stream carry copies and a few small kernels (filters, position order, unpacking; export, packing) that the host WAITS for beside the
      // persistent kernels of the alignment streams: highest priority, so that they get the first wave slots that come free (HLALA_IO_PRIORITY=normal: as before round 6)
      int prIo = prHigh; if(const char* e = getenv("HLALA_IO_PRIORITY")) { if(!strcmp(e, "normal")) prIo = (prLow + prHigh) / 2; else if(!strcmp(e, "low")) prIo = prLow; }
      if(hipStreamCreateWithPriority(&c->up, hipStreamNonBlocking, prIo) != hipSuccess || hipStreamCreateWithPriority(&c->rs, hipStreamNonBlocking, prIo) != hipSuccess) { c->err = "hipStreamCreate failed"; return fail(HLALA_E_DEVICE); } }
    { int prLow = 0, prHigh = 0; (void)hipDeviceGetStreamPriorityRange(&prLow, &prHigh);       // (numerically greatest = lowest priority)
      int pr = prLow; if(const char* e = getenv("HLALA_SIDE_PRIORITY")) { if(!strcmp(e, "high")) pr = prHigh; else if(!strcmp(e, "normal")) pr = (prLow + prHigh) / 2; }      // (tools/gpu_side_prio.sh)
      if(hipStreamCreateWithPriority(&c->side, hipStreamNonBlocking, pr) != hipSuccess) { c->err = "hipStreamCreate failed"; return fail(HLALA_E_DEVICE); } }
    if(hipStreamSynchronize(c->active) != hipSuccess) { c->err = "upload failed"; return fail(HLALA_E_DEVICE); }
    { std::lock_guard<std::mutex> g(g_ctx_mu); g_ctxs.push_back(c); }
    *out = c;
    return HLALA_OK;
}

void hlala_destroy(hlala_ctx* c)
{
    if(!c) return;
    DEV_GUARD(c);
    { std::lock_guard<std::mutex> g(g_ctx_mu); g_ctxs.erase(std::remove(g_ctxs.begin(), g_ctxs.end(), c), g_ctxs.end()); }
    if(!c->tail.empty()) (void)flush_tail(c);
    for(hlala_batch* b : c->batches) b->ctx = nullptr;       // a batch that outlives its context frees its own buffers
    for(void* p : c->allocs) if(p) (void)hipFree(p);
    if(c->create_scratch) { if(c->create_scratch_pinned) (void)hipHostFree(c->create_scratch); else free(c->create_scratch); }
    for(hlala_ctx::KeptReads& kr : c->kept) { (void)hipFree(kr.store); (void)hipFree(kr.start); (void)hipFree(kr.length); }
    for(auto& kv : c->pool) (void)hipFree(kv.second);
    if(c->side) { (void)hipStreamSynchronize(c->side); (void)hipStreamDestroy(c->side); }
    if(c->up) { (void)hipStreamSynchronize(c->up); (void)hipStreamDestroy(c->up); }
    if(c->rs) { (void)hipStreamSynchronize(c->rs); (void)hipStreamDestroy(c->rs); }
    if(c->evSideTail) (void)hipEventDestroy(c->evSideTail);
    delete c;
}

int hlala_graph_get_info(const hlala_ctx* c, hlala_graph_info* info)
{
    if(!c || !info) return HLALA_E_ARG;
    const FlatGraph& F = c->F;
    memset(info, 0, sizeof(*info));
    info->n_levels = F.L; info->n_nodes = F.N; info->n_edges = F.E; info->n_paths = (int)F.path_len.size();
    info->n_jump_entries = (int64_t)F.jf_node.size(); info->n_path_edges = (int64_t)F.path_edges.size();
    info->n_levelpos_entries = (int64_t)F.lp_seqid.size();
    info->max_nodes_per_level = F.max_nodes_per_level; info->max_out_degree = F.max_out_degree; info->max_in_degree = F.max_in_degree; info->max_jumps = F.max_jumps; info->max_parallel = F.max_parallel;
    for(uint8_t b : F.gap_stretch) info->n_gap_stretch_levels += b;
    return HLALA_OK;
}
int hlala_graph_get_nodes(const hlala_ctx* c, int32_t* node_orig, int32_t* level_off)
{
    if(!c) return HLALA_E_ARG;
    if(node_orig) memcpy(node_orig, c->F.node_orig.data(), c->F.node_orig.size() * 4);
    if(level_off) memcpy(level_off, c->F.level_off.data(), c->F.level_off.size() * 4);
    return HLALA_OK;
}
int hlala_graph_get_paths(const hlala_ctx* c, int32_t* first_node, int32_t* last_node, int32_t* length)
{
    if(!c) return HLALA_E_ARG;
    const FlatGraph& F = c->F;
    for(size_t p = 0; p < F.path_len.size(); p++) {
        if(first_node) first_node[p] = F.node_orig[F.path_first[p]];
        if(last_node) last_node[p] = F.node_orig[F.path_last[p]];
        if(length) length[p] = F.path_len[p];
    }
    return HLALA_OK;
}
int hlala_graph_get_gap_stretch(const hlala_ctx* c, uint8_t* in_stretch)
{
    if(!c || !in_stretch) return HLALA_E_ARG;
    memcpy(in_stretch, c->F.gap_stretch.data(), c->F.gap_stretch.size());
    return HLALA_OK;
}

// ------------------------------------------------------------------------------------ batches

static int batch_alloc_outputs(hlala_ctx* c, hlala_batch* b)
{
    DevBatch& B = b->B;
    size_t nc = (size_t)B.n_chains, nr = (size_t)B.n_reads, np = (size_t)B.n_pairs;
    // column rows: one per chain that passed the filters when they ran at creation (batch.h: chain_row), else one per chain
    if(!b->prepared) { B.n_rows = B.n_chains; B.chain_row = nullptr; }
    // (the column arrays are sized for the row count rounded up to a 64th of the chain count: consecutive batches of a sample differ by a per cent in the chains that
    //  pass the filters, and with exact sizes their arrays fell into different size classes of the pool every few batches -- a hipMalloc / hipFree pair of gigabytes, each a
    //  device-wide synchronisation, in front of the next launch)
    size_t rowsAlloc = (size_t)(B.n_rows > 0 ? B.n_rows : 1);
    if(b->prepared && nc >= 4096) { const size_t gran = nc / 64; rowsAlloc = (rowsAlloc + gran - 1) / gran * gran; if(rowsAlloc > nc) rowsAlloc = nc; }
    const size_t cs = rowsAlloc * (size_t)B.stride;
    int rc = 0;
#define AL(field, n, zero) do { rc = dev_alloc(c, b->allocs, (n), &B.field, zero); if(rc) return rc; } while(0)
    if(!b->prepared) { AL(seed_status, nc, true); AL(seed_ncols, nc, true); }
    AL(seed_begin, nc, true); AL(seed_end, nc, true); AL(seed_removed, nc, true);
    AL(seed_level, cs, false); AL(seed_edge, cs, false); AL(seed_g, cs, false); AL(seed_s, cs, false);
    AL(ext_status, nc, true); AL(ext_ncols, nc, true); AL(ext_begin, nc, true); AL(ext_end, nc, true); AL(ext_ll, nc, true);
    AL(dp_iters, 2 * nc, true); AL(dp_score, 2 * nc, true); AL(dp_ncols, 2 * nc, true); AL(dp_sb, 2 * nc, true); AL(dp_se, 2 * nc, true); AL(dp_err, 2 * nc, true);
    AL(dp_alias_head, 2 * nc, false); AL(dp_alias_next, 2 * nc, false);
    AL(ext_level, cs, false); AL(ext_edge, cs, false); AL(ext_g, cs, false); AL(ext_s, cs, false); AL(ext_fromseed, cs, false);
    AL(ext_firstlast, 4 * nc, true);
    AL(pair_status, np, true); AL(best_chain, nr, true); AL(n_comb, np, true); AL(pair_ll, np, true); AL(pair_mapq, np, true);
    AL(mate_mapq, nr, true); AL(strands_valid, np, true); AL(sel_mapq, nr * (size_t)B.stride, true);
    AL(pair_deferred, np, true); AL(pair_multi, 4 * np, false); AL(counters, 32, true); AL(work_counter, WC_N, true); AL(retry_list, 16 * nc, false);
    { DpItem* it = nullptr; rc = dev_alloc(c, b->allocs, 2 * nc, &it, false); if(rc) return rc; B.dp_items = it; }
    B.dp_nblk = (int)((nc + 255) / 256); if(B.dp_nblk < 1) B.dp_nblk = 1;
    B.dp_jf = c->jf_grid > 0 ? c->jf_margin + 1 : 0;
    B.dp_band = c->band_grid > 0 ? c->band_margin + 1 : 0;
    B.dp_band2 = c->band2_grid > 0 ? c->band2_margin + 1 : 0;
    B.dp_band2_maxj = c->band2_maxj;
    B.dp_band_risky = c->band_risky ? 1 : 0;
    AL(dp_blk, (size_t)DPL_N * B.dp_nblk + 1, false); AL(dp_list, 2 * nc, false);
    if(!b->prepared) {
        B.chain_order = nullptr; B.chain_bucket = nullptr; B.order_hist = nullptr; B.order_shift = c->order_shift; B.order_nb = c->order_nb; B.order_cost = c->order_cost; B.long_chunk_nodes = c->long_chunk_nodes; B.long_max_segs = c->long_max_segs;
        if(c->order_nb > 0 && !B.from_seeds && nc > 0) { AL(chain_order, nc, false); AL(chain_bucket, nc, false); AL(order_hist, (size_t)c->order_nb + 1, false); }
    }
    B.dbg = c->dbg_host;
#undef AL
    return 0;
}

static int batch_create_impl(hlala_ctx* c, const hlala_batch_in* in, hlala_batch** out, bool unpaired);
int hlala_batch_create(hlala_ctx* c, const hlala_batch_in* in, hlala_batch** out) { return batch_create_impl(c, in, out, false); }
int hlala_batch_create_unpaired(hlala_ctx* c, const hlala_batch_in* in, hlala_batch** out) { return batch_create_impl(c, in, out, true); }

static int batch_create_impl(hlala_ctx* c, const hlala_batch_in* in, hlala_batch** out, bool unpaired)
{
    DEV_GUARD(c);
    if(!c || !in || !out) return HLALA_E_ARG;
    UploadScope upscope_(c);           // uploads and the zeroing of the outputs run on the upload stream: beside the alignment of an earlier batch
    *out = nullptr;
    if(in->n_pairs < 0 || in->n_chains < 0) { c->err = "negative batch sizes"; return HLALA_E_ARG; }
    if(!c->d_contig_off) { c->err = "hlala_batch_create needs contigs (hlala_create was called without them)"; return HLALA_E_STATE; }
    static const bool hostTiming = getenv("HLALA_HOST_TIMING") != nullptr;       // (stderr: where hlala_batch_create spends the host thread's time)
    const auto tc0 = std::chrono::steady_clock::now(); auto tc1 = tc0, tc2 = tc0;
    hlala_batch* b = new hlala_batch(); b->ctx = c; c->batches.insert(b);
    DevBatch& B = b->B;
    B.n_pairs = in->n_pairs; B.n_reads = (unpaired ? 1 : 2) * in->n_pairs; B.n_chains = in->n_chains; B.stride = c->params.max_columns; B.from_seeds = 0; B.unpaired = unpaired ? 1 : 0;
    int nr = B.n_reads, nc = B.n_chains;
    auto fail = [&](int rc) { (void)hipStreamSynchronize(c->active); hlala_batch_destroy(b); return rc; };      // (uploads already queued -- from the caller's arrays, from the context's scratch -- end before their blocks go back to the pool)
    // the batch is a window into the caller's arrays (include/hlala_gpu.h: hlala_batch_in): 64-bit offsets that need not start at 0 are rebased here
    const int64_t rb0 = nr > 0 ? in->read_off[0] : 0, cb0 = nr > 0 ? in->chain_off[0] : 0;
    if(nr > 0) {
        const int64_t nbases64 = in->read_off[nr] - rb0, nc64 = in->chain_off[nr] - cb0;
        if(nbases64 < 0 || nc64 != (int64_t)nc || cb0 < 0 || rb0 < 0) { c->err = "inconsistent batch offsets"; return fail(HLALA_E_ARG); }
        if(nbases64 > 0x7FFFFFFFll) { c->err = "batch of " + std::to_string(nbases64) + " read bases: one batch holds at most 2^31 - 1 (cut the sample into more batches: hlala_seed_batch_window)"; return fail(HLALA_E_CAPACITY); }
    } else if(nc != 0) { c->err = "inconsistent batch offsets"; return fail(HLALA_E_ARG); }
    const int64_t gb0 = nc > 0 ? in->cigar_off[cb0] : 0;
    if(nc > 0) { const int64_t ng64 = in->cigar_off[cb0 + nc] - gb0; if(ng64 < 0) { c->err = "inconsistent batch offsets"; return fail(HLALA_E_ARG); }
                 if(ng64 > 0x7FFFFFFFll) { c->err = "batch of " + std::to_string(ng64) + " CIGAR operations: one batch holds at most 2^31 - 1"; return fail(HLALA_E_CAPACITY); } }
    b->first_chain = (uint32_t)cb0;
    // validation the reference would assert on, and the 32-bit arrays of the window (offsets rebased, the read of every chain).  Round 6: two fused passes -- one over the reads, one over
    // the chains -- on a few host threads into page-locked scratch of the context (five passes on the calling thread into fresh vectors before: 30 ms of hlala_batch_create's 75 for a
    // 1 M-pair batch, and their uploads from pageable memory were staged copies the caller waited for).  The FIRST violation in the order of the old passes is the one reported.
    int* chain_read = nullptr; int* read_off32 = nullptr; int* chain_off32 = nullptr; int* primary32 = nullptr; int* cigar_off32 = nullptr;
    {
        const size_t need = ((size_t)nc + (size_t)nr + 1 + (size_t)nr + 1 + (size_t)nr + (size_t)nc + 1 + 16) * sizeof(int);
        if(need > c->create_scratch_bytes) {
            if(c->create_scratch) { if(c->create_scratch_pinned) (void)hipHostFree(c->create_scratch); else free(c->create_scratch); c->create_scratch = nullptr; c->create_scratch_bytes = 0; }
            const size_t want = need + need / 8;
            void* p = nullptr;
            if(hipHostMalloc(&p, want, hipHostMallocDefault) == hipSuccess) c->create_scratch_pinned = true;
            else { (void)hipGetLastError(); p = malloc(want); c->create_scratch_pinned = false; }
            if(!p) { c->err = "out of host memory (batch scratch)"; return fail(HLALA_E_ARG); }
            c->create_scratch = p; c->create_scratch_bytes = want;
        }
        int* q = (int*)c->create_scratch;
        chain_read = q; q += nc; read_off32 = q; q += (size_t)nr + 1; chain_off32 = q; q += (size_t)nr + 1; primary32 = q; q += nr; cigar_off32 = q;
    }
    const int32_t* w_contig = in->chain_contig + cb0; const int32_t* w_pos = in->chain_pos + cb0; const int32_t* w_offset = in->chain_offset + cb0; const int32_t* w_as = in->chain_as + cb0;
    const uint8_t* w_rev = in->chain_reverse + cb0;
    {
        // violations by the pass that used to find them: 0 offsets of a read not ascending, 1 per-read checks (the smallest read decides, then the order of the checks within it),
        // 2 CIGAR offsets not ascending, 3 contig out of range, 4 paired read too long
        struct Viol { long long at[5]; int kind1; };
        const int nt = (nr + nc) >= (1 << 18) ? 4 : 1;
        std::vector<Viol> viol((size_t)nt);
        const int n_contigs = c->n_contigs;
        auto work = [&](int t) {
            Viol v; for(int i = 0; i < 5; i++) v.at[i] = -1; v.kind1 = 0;
            const int r0 = (int)((long long)nr * t / nt), r1 = (int)((long long)nr * (t + 1) / nt);
            for(int r = r0; r < r1; r++) {
                const int64_t ro0 = in->read_off[r] - rb0, ro1 = in->read_off[r + 1] - rb0, co0 = in->chain_off[r] - cb0, co1 = in->chain_off[r + 1] - cb0;
                if((ro1 < ro0 || co1 < co0) && v.at[0] < 0) v.at[0] = r;
                read_off32[r] = (int)ro0; chain_off32[r] = (int)co0;
                int kind = 0;
                if((int)co1 <= (int)co0) kind = 1;                         // read without alignments
                else if((int)ro1 < (int)ro0) kind = 2;                     // inconsistent batch offsets
                const int64_t pr = (int64_t)in->read_primary[r] - cb0;
                if(!kind && (pr < (int)co0 || pr >= (int)co1)) kind = 3;    // read_primary outside the read's chains
                if(kind && v.at[1] < 0) { v.at[1] = r; v.kind1 = kind; }
                primary32[r] = (int)pr;
                if(co0 >= 0 && co1 <= (int64_t)nc) for(int64_t k = co0; k < co1; k++) chain_read[k] = r;
                if(!unpaired && (int)ro1 - (int)ro0 > DP_SEQCAP && v.at[4] < 0) v.at[4] = r;
            }
            const int k0 = (int)((long long)nc * t / nt), k1 = (int)((long long)nc * (t + 1) / nt);
            for(int k = k0; k < k1; k++) {
                if(in->cigar_off[cb0 + k + 1] < in->cigar_off[cb0 + k] && v.at[2] < 0) v.at[2] = k;
                cigar_off32[k] = (int)(in->cigar_off[cb0 + k] - gb0);
                if((w_contig[k] < 0 || w_contig[k] >= n_contigs) && v.at[3] < 0) v.at[3] = k;
            }
            viol[(size_t)t] = v;
        };
        if(nt == 1) work(0);
        else {
            // (no exception may leave a C entry point: a thread that cannot be started -- a process at its thread limit -- leaves its share to the calling thread)
            std::vector<std::thread> th; int started = 0;
            try { th.reserve((size_t)nt); for(int t = 1; t < nt; t++) { th.emplace_back(work, t); started++; } } catch(...) { }
            work(0);
            for(int t = started + 1; t < nt; t++) work(t);
            for(std::thread& x : th) x.join();
        }
        if(nr > 0) { read_off32[nr] = (int)(in->read_off[nr] - rb0); chain_off32[nr] = (int)(in->chain_off[nr] - cb0); }
        if(nc > 0) cigar_off32[nc] = (int)(in->cigar_off[cb0 + nc] - gb0);
        long long first[5] = {-1, -1, -1, -1, -1}; int kind1 = 0;
        for(int t = 0; t < nt; t++) for(int i = 0; i < 5; i++) if(viol[(size_t)t].at[i] >= 0 && (first[i] < 0 || viol[(size_t)t].at[i] < first[i])) { first[i] = viol[(size_t)t].at[i]; if(i == 1) kind1 = viol[(size_t)t].kind1; }
        if(first[0] >= 0) { c->err = "inconsistent batch offsets"; return fail(HLALA_E_ARG); }
        if(first[1] >= 0) { c->err = kind1 == 1 ? "read without alignments" : (kind1 == 2 ? "inconsistent batch offsets" : "read_primary outside the read's chains"); return fail(HLALA_E_ARG); }
        if(first[2] >= 0) { c->err = "inconsistent batch offsets"; return fail(HLALA_E_ARG); }
        if(first[3] >= 0) { c->err = "chain_contig out of range"; return fail(HLALA_E_ARG); }
        // the extension DP of paired reads keys its cells with a 12-bit read coordinate (kernel_dp.hip: mk_key): longer reads belong in an unpaired batch
        if(first[4] >= 0) {
            const int r = (int)first[4];
            c->err = "paired read of " + std::to_string(read_off32[r + 1] - read_off32[r]) + " bases: the paired path holds reads of at most " + std::to_string(DP_SEQCAP) + " (use hlala_batch_create_unpaired for long reads)";
            return fail(HLALA_E_CAPACITY);
        }
    }
    size_t nbases = nr ? (size_t)read_off32[nr] : 0, ncig = nc ? (size_t)cigar_off32[nc] : 0;
    int rc = 0;
    tc1 = std::chrono::steady_clock::now();
#define UPB(field, ptr, n) do { rc = dev_upload(c, b->allocs, (ptr), (n), (std::remove_const<std::remove_pointer<decltype(B.field)>::type>::type**)&B.field); if(rc) return fail(rc); } while(0)
    UPB(read_off, read_off32, (size_t)nr + 1); UPB(read_quals, in->read_quals + rb0, nbases);
    if(in->read_bases_packed && nr > 0) {
        // 4-bit packed bases (half the upload): read R of the sample starts at byte (base offset + R + 1) >> 1; unpacked on the device into the array the kernels read
        const int64_t R0 = in->first_read;
        if(R0 < 0) { c->err = "first_read is negative"; return fail(HLALA_E_ARG); }
        const int64_t p0 = (rb0 + R0 + 1) >> 1, p1 = ((in->read_off[nr] + R0 + (int64_t)nr + 1) >> 1) + 1;
        uint8_t* dPacked = nullptr; uint8_t* dBases = nullptr;
        rc = dev_upload(c, b->allocs, in->read_bases_packed + p0, (size_t)(p1 - p0), &dPacked); if(rc) return fail(rc);
        rc = dev_alloc(c, b->allocs, nbases, &dBases, false); if(rc) return fail(rc);
        hipLaunchKernelGGL(k_unpack_bases, dim3((unsigned)nr), dim3(64), 0, c->active, (const uint8_t*)dPacked, B.read_off, nr, (long long)rb0, (long long)R0, (long long)p0, dBases);
        { hipError_t el = hipGetLastError(); if(el != hipSuccess) { c->err = std::string("k_unpack_bases: ") + hipGetErrorString(el); return fail(HLALA_E_DEVICE); } }
        B.read_bases = dBases;
    } else {
        if(!in->read_bases && nbases > 0) { c->err = "neither read_bases nor read_bases_packed"; return fail(HLALA_E_ARG); }
        UPB(read_bases, in->read_bases + rb0, nbases);
    }
    UPB(chain_off, chain_off32, (size_t)nr + 1); UPB(read_primary, primary32, (size_t)nr);
    UPB(chain_read, chain_read, (size_t)nc);
    UPB(chain_contig, w_contig, (size_t)nc); UPB(chain_pos, w_pos, (size_t)nc); UPB(chain_offset, w_offset, (size_t)nc);
    UPB(chain_as, w_as, (size_t)nc); UPB(chain_reverse, w_rev, (size_t)nc);
    UPB(cigar_off, cigar_off32, (size_t)nc + 1); UPB(cigar, in->cigar + gb0, ncig);
#undef UPB
    // The OUTPUT arrays are allocated by the first stage call (ensure_outputs): a caller that uploads batch i+2 while batches i and i+1 are aligned and read back
    // (two alignments in flight, one upload ahead) then holds the inputs of three batches -- 0.9 GB each -- but the outputs of two, and the outputs of the batch it
    // destroys are the pool blocks the next alignment takes.  Until then the device descriptor holds the inputs only (what hlala_kmer_presence reads).
    // ... except the few per-chain arrays of the FILTERS and the POSITION ORDER, which read inputs only and run here, on the upload stream: the number of chains that
    // passed -- a third of them on an MHC-scale graph -- comes back with the synchronisation this function ends on anyway, and the column arrays (20 bytes per column slot:
    // 47 GB for the 6.1 M chains of a 1 M-pair batch) are then sized for those chains alone (batch.h: chain_row; HLALA_ROWS_ALL=1: a row per chain, filters at the stage call)
    if(c->order_nb > 0 && nc > 0 && !c->rows_all) {
#define AL(field, n, zero) do { rc = dev_alloc(c, b->allocs, (n), &B.field, zero); if(rc) return fail(rc); } while(0)
        AL(seed_status, (size_t)nc, true); AL(seed_ncols, (size_t)nc, true);
        AL(chain_order, (size_t)nc, false); AL(chain_bucket, (size_t)nc, false); AL(chain_row, (size_t)nc, false); AL(order_hist, (size_t)c->order_nb + 1, true);
#undef AL
        B.order_shift = c->order_shift; B.order_nb = c->order_nb; B.order_cost = c->order_cost; B.long_chunk_nodes = c->long_chunk_nodes; B.long_max_segs = c->long_max_segs;
        b->prepared = true;
    }
    rc = dev_upload(c, b->allocs, &b->B, 1, &b->dB); if(rc) return fail(rc);
    if(b->prepared) {
        hipLaunchKernelGGL(k_filter_chains, dim3((nr + 255) / 256), dim3(256), 0, c->active, c->dG, b->dB, c->d_contig_off, c->d_contig_level);
        hipLaunchKernelGGL(k_order_scan, dim3(1), dim3(ORDER_SCAN_THREADS), 0, c->active, B.order_hist, B.order_nb);
        hipLaunchKernelGGL(k_order_scatter, dim3((nc + 255) / 256), dim3(256), 0, c->active, b->dB);
        { hipError_t el = hipGetLastError(); if(el != hipSuccess) { c->err = std::string("k_order_scatter: ") + hipGetErrorString(el); return fail(HLALA_E_DEVICE); } }
        HIP_TRY_F(c, hipMemcpyAsync(&b->n_rows_host, B.order_hist + (B.order_nb - 1), sizeof(int), hipMemcpyDeviceToHost, c->active), fail);
    }
    tc2 = std::chrono::steady_clock::now();
    HIP_TRY_F(c, hipStreamSynchronize(c->active), fail);
    if(hostTiming) { const auto tc3 = std::chrono::steady_clock::now(); auto ms = [](std::chrono::steady_clock::duration d) { return std::chrono::duration<double, std::milli>(d).count(); };
                     fprintf(stderr, "hlala_batch_create: checks and rebasing %.1f ms, allocations + uploads + kernels queued %.1f ms, waiting for the upload stream %.1f ms\n", ms(tc1 - tc0), ms(tc2 - tc1), ms(tc3 - tc2)); }
    if(b->prepared) {
        if(b->n_rows_host < 0 || b->n_rows_host > nc) { c->err = "position order counted " + std::to_string(b->n_rows_host) + " chains of " + std::to_string(nc); return fail(HLALA_E_DEVICE); }
        B.n_rows = b->n_rows_host;        // (the device descriptor gets it with the output arrays: ensure_outputs)
    }
    *out = b;
    return HLALA_OK;
}

// output arrays of a batch that came through hlala_batch_create / _unpaired: allocated (pool) and zeroed on the stream of the stage call that needs them first, then the
// device descriptor is written again with their addresses
static int ensure_outputs(hlala_ctx* c, hlala_batch* b)
{
    if(b->outputs_ready) return HLALA_OK;
    int rc = batch_alloc_outputs(c, b); if(rc) return rc;
    HIP_TRY(c, hipMemcpyAsync(b->dB, &b->B, sizeof(DevBatch), hipMemcpyHostToDevice, c->active));
    b->outputs_ready = true;
    return HLALA_OK;
}

int hlala_batch_create_from_seeds(hlala_ctx* c, const hlala_seeds_in* in, hlala_batch** out)
{
    DEV_GUARD(c);
    if(!c || !in || !out) return HLALA_E_ARG;
    UploadScope upscope_(c);
    *out = nullptr;
    hlala_batch* b = new hlala_batch(); b->ctx = c; c->batches.insert(b);
    DevBatch& B = b->B;
    B.n_pairs = 0; B.n_reads = in->n_reads; B.n_chains = in->n_chains; B.stride = c->params.max_columns; B.from_seeds = 1;
    int nr = B.n_reads, nc = B.n_chains; int stride = B.stride;
    auto fail = [&](int rc) { hlala_batch_destroy(b); return rc; };
    size_t nbases = nr ? (size_t)in->read_off[nr] : 0;
    int rc = 0;
    rc = dev_upload(c, b->allocs, in->read_off, (size_t)nr + 1, (int**)&B.read_off); if(rc) return fail(rc);
    rc = dev_upload(c, b->allocs, in->read_bases, nbases, (uint8_t**)&B.read_bases); if(rc) return fail(rc);
    rc = dev_upload(c, b->allocs, in->read_quals, nbases, (uint8_t**)&B.read_quals); if(rc) return fail(rc);
    rc = dev_upload(c, b->allocs, in->chain_read, (size_t)nc, (int**)&B.chain_read); if(rc) return fail(rc);
    rc = dev_upload(c, b->allocs, in->chain_reverse, (size_t)nc, (uint8_t**)&B.chain_reverse); if(rc) return fail(rc);
    rc = batch_alloc_outputs(c, b); if(rc) return fail(rc);
    rc = dev_upload(c, b->allocs, &b->B, 1, &b->dB); if(rc) return fail(rc);
    // scatter the ragged seed columns into the fixed-stride layout
    std::vector<int> st((size_t)nc, HLALA_CHAIN_OK), ncols((size_t)nc), lev((size_t)nc * stride, -1), edg((size_t)nc * stride, -1);
    std::vector<uint8_t> g((size_t)nc * stride, 0), s((size_t)nc * stride, 0);
    for(int k = 0; k < nc; k++) {
        int n = in->col_off[k + 1] - in->col_off[k];
        if(in->chain_read[k] < 0 || in->chain_read[k] >= nr) { c->err = "chain_read out of range"; return fail(HLALA_E_ARG); }
        if(n > stride || n < 1) { st[k] = HLALA_CHAIN_ERR_COLUMNS; ncols[k] = 0; continue; }
        ncols[k] = n;
        for(int j = 0; j < n; j++) {
            size_t o = (size_t)k * stride + j; int i = in->col_off[k] + j;
            lev[o] = in->col_level[i]; edg[o] = in->col_edge[i]; g[o] = in->col_gchar[i]; s[o] = in->col_schar[i];
        }
    }
    HIP_TRY_F(c, hipMemcpyAsync(B.seed_status, st.data(), (size_t)nc * 4, hipMemcpyHostToDevice, c->active), fail);
    HIP_TRY_F(c, hipMemcpyAsync(B.seed_ncols, ncols.data(), (size_t)nc * 4, hipMemcpyHostToDevice, c->active), fail);
    HIP_TRY_F(c, hipMemcpyAsync(B.seed_begin, in->chain_seq_begin, (size_t)nc * 4, hipMemcpyHostToDevice, c->active), fail);
    HIP_TRY_F(c, hipMemcpyAsync(B.seed_end, in->chain_seq_end, (size_t)nc * 4, hipMemcpyHostToDevice, c->active), fail);
    HIP_TRY_F(c, hipMemcpyAsync(B.seed_level, lev.data(), lev.size() * 4, hipMemcpyHostToDevice, c->active), fail);
    HIP_TRY_F(c, hipMemcpyAsync(B.seed_edge, edg.data(), edg.size() * 4, hipMemcpyHostToDevice, c->active), fail);
    HIP_TRY_F(c, hipMemcpyAsync(B.seed_g, g.data(), g.size(), hipMemcpyHostToDevice, c->active), fail);
    HIP_TRY_F(c, hipMemcpyAsync(B.seed_s, s.data(), s.size(), hipMemcpyHostToDevice, c->active), fail);
    HIP_TRY_F(c, hipStreamSynchronize(c->active), fail);
    b->staged = 1; b->outputs_ready = true;
    *out = b;
    return HLALA_OK;
}

int hlala_batch_set_first_chain(hlala_batch* b, uint32_t first_chain)
{
    if(!b) return HLALA_E_ARG;
    b->first_chain = first_chain;
    return HLALA_OK;
}

void hlala_batch_destroy(hlala_batch* b)
{
    if(!b) return;
    hlala_ctx* c = b->ctx;
    DEV_GUARD(c);
    if(c) {
        if(b->tail_pooled) (void)flush_tail(c);       // (its pending classes run with whatever the pool holds; should that fail, the batch leaves the pool below)
        c->tail.erase(std::remove(c->tail.begin(), c->tail.end(), b), c->tail.end());
        c->batches.erase(b);
        if(b->side_inflight) (void)hipEventSynchronize(b->evDone);
        if(b->mainValid) (void)hipEventSynchronize(b->evMain);       // nothing of this batch may still be running when its buffers are handed to the next one (readers and uploads return synchronised)
        for(void* p : b->allocs) pool_release(c, p);
    } else for(void* p : b->allocs) if(p) (void)hipFree(p);
    if(b->evDone) (void)hipEventDestroy(b->evDone);
    if(b->evMain) (void)hipEventDestroy(b->evMain);
    for(int i = 0; i < 14; i++) { if(b->ev[i]) (void)hipEventDestroy(b->ev[i]); if(b->evC[i / 2][i % 2]) (void)hipEventDestroy(b->evC[i / 2][i % 2]); }
    for(int i = 0; i < 8; i++) if(b->evSide[i]) (void)hipEventDestroy(b->evSide[i]);
    if(b->evJF) (void)hipEventDestroy(b->evJF);
    for(int i = 0; i < 2; i++) if(b->evBand[i]) (void)hipEventDestroy(b->evBand[i]);
    for(int i = 0; i < 2; i++) if(b->evBand2[i]) (void)hipEventDestroy(b->evBand2[i]);
    delete b;
}

static int check_launch(hlala_ctx* c, const char* what)
{
    hipError_t e = hipGetLastError();
    if(e != hipSuccess) { c->err = std::string(what) + ": " + hipGetErrorString(e); return HLALA_E_DEVICE; }
    return 0;
}

int hlala_project_chains(hlala_ctx* c, hlala_batch* b)
{
    DEV_GUARD(c);
    if(!c || !b) return HLALA_E_ARG;
    { int rj = join_side(c, b); if(rj) return rj; }
    if(b->B.from_seeds) { c->err = "batch was created from seeds: stage A not available"; return HLALA_E_STATE; }
    { int re = batch_events(c, b); if(re) return re; }
    { int ro = ensure_outputs(c, b); if(ro) return ro; }
    DevBatch& B = b->B;
    HIP_TRY(c, hipMemsetAsync(B.work_counter, 0, WC_N * sizeof(int), c->active));
    HIP_TRY(c, hipMemsetAsync(B.counters, 0, 32 * sizeof(u64), c->active));
    HIP_TRY(c, hipEventRecord(b->ev[0], c->active));
    if(B.n_chains > 0) {
        int threads = 256, blocks = (B.n_reads + threads - 1) / threads;
        // (the filters and the position order of a batch ran when it was created, hlala_batch_create: they read inputs only)
        if(!b->prepared && B.order_hist) HIP_TRY(c, hipMemsetAsync(B.order_hist, 0, ((size_t)B.order_nb + 1) * sizeof(int), c->active));
        if(!b->prepared) hipLaunchKernelGGL(k_filter_chains, dim3(blocks), dim3(threads), 0, c->active, c->dG, b->dB, c->d_contig_off, c->d_contig_level);
        if(!b->prepared && B.order_hist) {
            // chains into position order (kernel_order.hip): every later kernel that walks the graph takes them from B.chain_order
            hipLaunchKernelGGL(k_order_scan, dim3(1), dim3(ORDER_SCAN_THREADS), 0, c->active, B.order_hist, B.order_nb);
            hipLaunchKernelGGL(k_order_scatter, dim3((B.n_chains + 255) / 256), dim3(256), 0, c->active, b->dB);
            int rco = check_launch(c, "k_order_scatter"); if(rco) return rco;
        }
        int grid = B.n_chains < c->proj_grid ? B.n_chains : c->proj_grid;
        if(c->proj_long_slabs)
            hipLaunchKernelGGL((k_project_chains<ProjLdsLong>), dim3(grid), dim3(64), 0, c->active, c->dG, b->dB, c->d_contig_off, c->d_contig_seq, c->d_contig_level,
                               c->proj_slabs, c->proj_slab_bytes, c->proj_long_slabs, c->proj_long_slab_bytes, c->proj_long_stagger);      // (long layout: the last argument staggers the wavefronts' starts)
        else
            if(c->params.max_columns <= PROJ_CAP_SHORT)
                hipLaunchKernelGGL((k_project_chains<ProjLdsShort>), dim3(grid), dim3(64), 0, c->active, c->dG, b->dB, c->d_contig_off, c->d_contig_seq, c->d_contig_level,
                                   c->proj_slabs, c->proj_slab_bytes, (char*)nullptr, (size_t)0, c->rethread_slabs ? 1 : 0);
            else
                hipLaunchKernelGGL((k_project_chains<ProjLds>), dim3(grid), dim3(64), 0, c->active, c->dG, b->dB, c->d_contig_off, c->d_contig_seq, c->d_contig_level,
                               c->proj_slabs, c->proj_slab_bytes, (char*)nullptr, (size_t)0, c->rethread_slabs ? 1 : 0);
        int rc = check_launch(c, "k_project_chains"); if(rc) return rc;
        if(c->rethread_slabs) {
            hipLaunchKernelGGL(k_rethread_chains, dim3(c->rethread_grid), dim3(64), 0, c->active, c->dG, b->dB, c->rethread_slabs, c->rethread_slab_bytes);
            rc = check_launch(c, "k_rethread_chains"); if(rc) return rc;
        }
    }
    HIP_TRY(c, hipEventRecord(b->ev[1], c->active));
    b->staged |= 1;
    return mark_main(c, b);
}

static int extend_impl(hlala_ctx* c, hlala_batch* b, bool fused, int phase = 0);
int hlala_extend_chains(hlala_ctx* c, hlala_batch* b) { return extend_impl(c, b, false); }

// fused = called from hlala_align_batch on a paired batch.  The 16- / 32- / 64-lane classes hold all but a few percent of the DP calls; the rest (wide,
// broad, large, in-memory) are few, long calls that cannot fill the chip.  Fused, they run on the side stream, followed there by a second stitch /
// pairing pass over the pairs that own them (pair_deferred, set by the DP kernels when they hand an item to one of these classes), while the main
// stream stitches and pairs everything else and is then free for the caller's next batch.  b->evDone orders later users of the batch behind the side work.
// phase (fused only; hlala_align_batch): 0 = the whole stage, the side-stream classes forked off as soon as the 64-lane class is done; 1 = the main-stream part alone (classes
// before DP_SIDE_TIER, first stitch pass), 2 = the side-stream part alone (the later classes, second stitch pass) -- queued by hlala_align_batch AFTER the main stream's
// pairing pass: k_pair_chains is a short, latency-bound kernel that needs 6 KB of LDS per wavefront, and beside the wide class (seven blocks of 22 KB per CU: 154 of a
// CU's 160 KB) hardly a block of it fits: 20.9 ms for a pairing pass that takes 4.1 ms alone (profiles/r05_experiments.txt, 19).  Not the default: the next batch's
// projection (11.7 KB per block) then meets the wide class instead.
static int extend_impl(hlala_ctx* c, hlala_batch* b, bool fused, int phase)
{
    DEV_GUARD(c);
    if(!c || !b) return HLALA_E_ARG;
    if(!(b->staged & 1)) { c->err = "hlala_extend_chains before seed chains exist"; return HLALA_E_STATE; }
    // phases 3 / 4 (round 6, tail pool): 3 = the main-stream part + the first side-stream class (wide) of THIS batch, the later classes left pending; 4 = the second stitch
    // pass alone, queued by flush_tail behind the pooled launches of those classes
    const bool sidePartOnly = phase == 2 || phase == 4;
    if(!sidePartOnly) { int rj = join_side(c, b); if(rj) return rj; }
    { int re = batch_events(c, b); if(re) return re; }
    DevBatch& B = b->B;
    if(B.unpaired || B.from_seeds || B.n_pairs <= 0 || B.n_chains <= 0) { fused = false; if(sidePartOnly) return HLALA_OK; phase = 0; }
    const bool mainPart = !sidePartOnly;
    if(mainPart) {
    b->side_used = false; b->side_pending = false;
    if(B.n_pairs > 0) HIP_TRY(c, hipMemsetAsync(B.pair_deferred, 0, (size_t)B.n_pairs, c->active));
    HIP_TRY(c, hipMemsetAsync(B.work_counter + 1, 0, sizeof(int), c->active));
    HIP_TRY(c, hipMemsetAsync(B.work_counter + 4, 0, (WC_N - 4) * sizeof(int), c->active));       // [4..6] jump-free lists, [7..] stitch, DP items, retry lists, band / fail-over lists
    if(B.from_seeds) HIP_TRY(c, hipMemsetAsync(B.counters, 0, 32 * sizeof(u64), c->active));
    HIP_TRY(c, hipEventRecord(b->ev[2], c->active));
    }
    if(B.n_chains > 0) {
        DpItem* items = (DpItem*)B.dp_items;
        const u32 seed = c->params.rng_seed + 2u * b->first_chain;
        int rc = 0;
        if(mainPart) {
        HIP_TRY(c, hipMemsetAsync(B.dp_alias_head, 0xFF, (size_t)2 * B.n_chains * sizeof(int), c->active));       // -1: k_dp_items links the duplicates of a DP to it
        HIP_TRY(c, hipMemsetAsync(B.dp_alias_next, 0xFF, (size_t)2 * B.n_chains * sizeof(int), c->active));
        // items, then the ten dense lists of the first classes (three band lists, jump-free, general; left / right each) in position order: counts per block, their scan, the slots
        HIP_TRY(c, hipMemsetAsync(B.dp_blk, 0, ((size_t)DPL_N * B.dp_nblk + 1) * sizeof(int), c->active));
        hipLaunchKernelGGL(k_dp_items, dim3(B.dp_nblk), dim3(256), 0, c->active, c->dG, b->dB, items);
        rc = check_launch(c, "k_dp_items"); if(rc) return rc;
        hipLaunchKernelGGL(k_order_scan, dim3(1), dim3(ORDER_SCAN_THREADS), 0, c->active, B.dp_blk, DPL_N * B.dp_nblk + 1);
        hipLaunchKernelGGL(k_dp_lists, dim3(B.dp_nblk), dim3(256), 0, c->active, b->dB, (const DpItem*)items);
        rc = check_launch(c, "k_dp_lists"); if(rc) return rc;
        }
        // every DP item first runs in the 16-lane class; the item count lives on the device, idle groups leave at once.
        // Items that outgrew it: two DPs per wave, then one wave per DP, then the classes with wider frontiers (fewer blocks per CU).
        // Each class is timed with its own pair of events on the stream it runs on (evC); ev[7] / ev[6] / ev[10] keep marking the start of the 16-lane
        // class, its end and the end of the 64-lane class on the main stream.
        hipStream_t ws = c->active;
        const DpPoolArgs noPool{};                                   // classes before DP_POOL_TIER take the batch of their plain arguments
        DpPoolArgs one{}; one.n = 1; one.seed[0] = seed; one.B[0] = b->dB; one.items[0] = items; one.bases[0] = (const uint8_t*)B.read_bases;      // the later ones a list of batches: this one
        auto run_class = [&](int tier) -> int {
            if(fused && tier == DP_SIDE_TIER) {
                if(!b->evDone) HIP_TRY(c, hipEventCreateWithFlags(&b->evDone, hipEventDisableTiming));
                HIP_TRY(c, hipEventRecord(b->evSide[0], c->active)); HIP_TRY(c, hipStreamWaitEvent(c->side, b->evSide[0], 0));
                ws = c->side;
                HIP_TRY(c, hipEventRecord(b->evSide[1], c->side));
            }
            // the slabs of the classes from DP_SIDE_TIER on belong to the context: a launch on the main stream (stage calls, unpaired and from-seeds batches)
            // goes behind whatever an earlier fused alignment of ANOTHER batch still has queued on the side stream
            if(!fused && tier == DP_SIDE_TIER && c->sideTailValid) HIP_TRY(c, hipStreamWaitEvent(c->stream, c->evSideTail, 0));
            HIP_TRY(c, hipEventRecord(b->evC[tier][0], ws));
            switch(tier) {
            case 0:
                // the calls that meet no gap-path jump in the instantiation without the early-cell machinery, then the others (same slabs: one after the other)
                if(c->jf_grid > 0) {
                    hipLaunchKernelGGL((k_dp<DpTinyJF, 0>), dim3(c->jf_grid), dim3(DpTinyJF::THREADS), 0, ws, c->dG, b->dB, items, c->tiny_slabs, c->tiny_slab_bytes, seed, c->G.nrec_out, c->G.nrec_in, B.read_bases, noPool);
                    int rcj = check_launch(c, "k_dp<DpTinyJF>"); if(rcj) return rcj;
                    HIP_TRY(c, hipEventRecord(b->evJF, ws));
                }
                hipLaunchKernelGGL((k_dp<DpTiny, 0>), dim3(c->tiny_grid), dim3(DpTiny::THREADS), 0, ws, c->dG, b->dB, items, c->tiny_slabs, c->tiny_slab_bytes, seed, c->G.nrec_out, c->G.nrec_in, B.read_bases, noPool); break;
            case 1: hipLaunchKernelGGL((k_dp<DpMid, 1>), dim3(c->mid_grid), dim3(DpMid::THREADS), 0, ws, c->dG, b->dB, items, c->mid_slabs, c->mid_slab_bytes, seed, c->G.nrec_out, c->G.nrec_in, B.read_bases, noPool); break;
            case 2: hipLaunchKernelGGL((k_dp<DpSmall, 2>), dim3(c->ext_grid), dim3(DpSmall::THREADS), 0, ws, c->dG, b->dB, items, c->ext_slabs, c->ext_slab_bytes, seed, c->G.nrec_out, c->G.nrec_in, B.read_bases, noPool); break;
            case 3: hipLaunchKernelGGL((k_dp<DpWide, 3>), dim3(c->wide_grid), dim3(DpWide::THREADS), 0, ws, c->dG, b->dB, items, c->wide_slabs, c->ext_slab_bytes, seed, c->G.nrec_out, c->G.nrec_in, B.read_bases, noPool); break;
            case 4: hipLaunchKernelGGL((k_dp<DpBroad, 4>), dim3(c->broad_grid), dim3(DpBroad::THREADS), 0, ws, c->dG, b->dB, items, c->large_slabs, c->large_slab_bytes, seed, c->G.nrec_out, c->G.nrec_in, B.read_bases, one); break;
            case 5: hipLaunchKernelGGL((k_dp<DpLarge, 5>), dim3(c->retry_grid), dim3(DpLarge::THREADS), 0, ws, c->dG, b->dB, items, c->large_slabs + c->large_slab_bytes * (size_t)c->broad_grid, c->large_slab_bytes, seed, c->G.nrec_out, c->G.nrec_in, B.read_bases, one); break;
            default: hipLaunchKernelGGL((k_dp<DpHuge, 6>), dim3(c->huge_grid), dim3(DpHuge::THREADS), 0, ws, c->dG, b->dB, items, c->huge_slabs, c->huge_slab_bytes, seed, c->G.nrec_out, c->G.nrec_in, B.read_bases, one); break;
            }
            int rc_ = check_launch(c, "k_dp"); if(rc_) return rc_;
            HIP_TRY(c, hipEventRecord(b->evC[tier][1], ws));
            return 0;
        };
        if(mainPart) {
        // calls on linear stretches of the graph first: anti-diagonals in registers, four calls per wavefront (kernel_dp_band.hip); what it cannot finish is on the
        // fail-over list the general 16-lane instantiation draws after its own
        b->band_used = c->band_grid > 0;
        if(b->band_used) {
            HIP_TRY(c, hipEventRecord(b->evBand[0], c->active));
            hipLaunchKernelGGL((k_dp_band<16>), dim3(c->band_grid), dim3(64), 0, c->active, c->dG, b->dB, (const DpItem*)items, seed, (const uint8_t*)B.read_bases, c->G.lin_label, c->G.lin_eid);
            hipLaunchKernelGGL((k_dp_band<32>), dim3(c->band_grid), dim3(64), 0, c->active, c->dG, b->dB, (const DpItem*)items, seed, (const uint8_t*)B.read_bases, c->G.lin_label, c->G.lin_eid);
            hipLaunchKernelGGL((k_dp_band<64>), dim3(c->band_grid), dim3(64), 0, c->active, c->dG, b->dB, (const DpItem*)items, seed, (const uint8_t*)B.read_bases, c->G.lin_label, c->G.lin_eid);
            rc = check_launch(c, "k_dp_band"); if(rc) return rc;
            HIP_TRY(c, hipEventRecord(b->evBand[1], c->active));
        }
        // calls beside gap stretches next: two tracks and one gap-path jump in registers (kernel_dp_band2.hip); same fail-over list
        b->band2_used = c->band2_grid > 0;
        if(b->band2_used) {
            HIP_TRY(c, hipEventRecord(b->evBand2[0], c->active));
            hipLaunchKernelGGL((k_dp_band2<16>), dim3(c->band2_grid), dim3(64), 0, c->active, c->dG, b->dB, (const DpItem*)items, seed, (const uint8_t*)B.read_bases, c->band2_slabs);
            hipLaunchKernelGGL((k_dp_band2<32>), dim3(c->band2_grid), dim3(64), 0, c->active, c->dG, b->dB, (const DpItem*)items, seed, (const uint8_t*)B.read_bases, c->band2_slabs);
            hipLaunchKernelGGL((k_dp_band2<64>), dim3(c->band2_grid), dim3(64), 0, c->active, c->dG, b->dB, (const DpItem*)items, seed, (const uint8_t*)B.read_bases, c->band2_slabs);
            rc = check_launch(c, "k_dp_band2"); if(rc) return rc;
            HIP_TRY(c, hipEventRecord(b->evBand2[1], c->active));
        }
        HIP_TRY(c, hipEventRecord(b->ev[7], c->active));
        rc = run_class(0); if(rc) return rc;
        HIP_TRY(c, hipEventRecord(b->ev[6], c->active));
        }
        for(int tier = 1; tier <= DP_LAST_TIER; tier++) {
            if(phase == 4) break;
            if(phase == 1 && tier >= DP_SIDE_TIER) break;
            if(phase == 2 && tier < DP_SIDE_TIER) continue;
            if(phase == 3 && tier >= DP_POOL_TIER) break;
            rc = run_class(tier); if(rc) return rc;
            if(tier == 2) HIP_TRY(c, hipEventRecord(b->ev[10], c->active));
        }
        if(!fused) HIP_TRY(c, hipEventRecord(b->ev[8], c->active));
        const int sgrid = B.n_chains < c->stitch_grid ? B.n_chains : c->stitch_grid;
        if(fused && phase == 1) b->side_pending = true;
        if(fused && phase == 3) { b->side_pending = true; b->side_used = true; }
        if(fused && phase != 1 && phase != 3) {
            // second pass (side): the chains of the deferred pairs, work counter 36; first pass (main): all the others, work counter 7
            { const int pgrid = (B.n_pairs + 63) / 64, cap = c->stitch_grid / 20;       // the second pass sweeps the pairs' flags, 64 per wave and round: one wave per CU finds room beside the next batch's persistent kernels
              hipLaunchKernelGGL(k_stitch_chains, dim3(pgrid < cap ? (pgrid > 0 ? pgrid : 1) : cap), dim3(64), 0, c->side, c->dG, c->dT, b->dB, (const uint8_t*)B.pair_deferred, 2, 0, 0); }
            rc = check_launch(c, "k_stitch_chains (side)"); if(rc) return rc;
            HIP_TRY(c, hipEventRecord(b->evDone, c->side));
            HIP_TRY(c, hipEventRecord(c->evSideTail, c->side)); c->sideTailValid = true;
            b->side_inflight = true; b->side_used = true; b->side_pending = true;
        }
        if(mainPart) {
        hipLaunchKernelGGL(k_stitch_chains, dim3(sgrid), dim3(64), 0, c->active, c->dG, c->dT, b->dB, (const uint8_t*)B.pair_deferred, fused ? 1 : 0, c->stitch_draw, c->stitch_by_row);
        rc = check_launch(c, "k_stitch_chains"); if(rc) return rc;
        }
    }
    if(sidePartOnly) return HLALA_OK;
    HIP_TRY(c, hipEventRecord(b->ev[3], c->active));
    b->staged |= 2;
    return mark_main(c, b);
}

static int pair_impl(hlala_ctx* c, hlala_batch* b, int phase);
int hlala_pair_chains(hlala_ctx* c, hlala_batch* b) { return pair_impl(c, b, 0); }
// phase: 0 = the whole stage; 1 = the main-stream pass alone, 2 = the side-stream pass alone (hlala_align_batch: extend_impl)
static int pair_impl(hlala_ctx* c, hlala_batch* b, int phase)
{
    DEV_GUARD(c);
    if(!c || !b) return HLALA_E_ARG;
    if(b->B.from_seeds) { c->err = "batch was created from seeds: stage C not available"; return HLALA_E_STATE; }
    if(!(b->staged & 2)) { c->err = "hlala_pair_chains before hlala_extend_chains"; return HLALA_E_STATE; }
    { int re = batch_events(c, b); if(re) return re; }
    DevBatch& B = b->B;
    const bool fused = b->side_pending;
    if(!fused) { if(phase == 2) return HLALA_OK; phase = 0; }
    if(!fused) { int rj = join_side(c, b); if(rj) return rj; }
    if(phase != 2) {
    HIP_TRY(c, hipMemsetAsync(B.work_counter + 2, 0, sizeof(int), c->active));
    HIP_TRY(c, hipEventRecord(b->ev[4], c->active));
    }
    if(B.n_pairs > 0) {
        const int grid = B.n_pairs < c->pair_grid ? B.n_pairs : c->pair_grid;
        // k_pair_chains finishes the pairs with one combination and lists the others, k_pair_multi<., false / true> runs the two lists (kernel_pair.hip); the main-
        // and the side-stream pass have their own lists, counters and combination scratch
        auto launch_pair = [&](hipStream_t st, int mode, int counterIdx) -> int {
            const int pass = st == c->side ? 1 : 0, multiBase = WC_PAIR_MULTI + 4 * pass;
            int* multiList = B.pair_multi + (size_t)pass * 2 * (size_t)B.n_pairs;
            double* scratch = c->pair_scratch + (pass ? (size_t)c->pair_grid * PAIR_COMB : 0);
            // (the lists' counters are among those the extension stage clears; a pairing stage called again on its own clears them here.  Not on the side stream: a fill
            //  kernel queued there waits 13-36 ms for a wave slot beside the persistent kernels -- profiles/r06_experiments.txt)
            if(!fused) HIP_TRY(c, hipMemsetAsync(B.work_counter + multiBase, 0, 4 * sizeof(int), st));
            const int lean = B.n_pairs < c->pair_lean_grid ? B.n_pairs : c->pair_lean_grid;
            const int g0 = mode == 2 ? (grid < c->pair_grid / 5 ? grid : c->pair_grid / 5) : grid, g1 = grid < c->pair_grid / 5 ? grid : c->pair_grid / 5;      // (the second pass and the general class hold a few thousand pairs at most)
            if(B.unpaired) {
                hipLaunchKernelGGL((k_pair_chains<true>), dim3(lean), dim3(64), 0, st, c->dG, c->dT, b->dB, (const uint8_t*)B.pair_deferred, mode, counterIdx, multiList, multiBase);
                hipLaunchKernelGGL((k_pair_multi<true, false>), dim3(g0), dim3(64), 0, st, c->dG, c->dT, b->dB, (const int*)multiList, multiBase, scratch);
                hipLaunchKernelGGL((k_pair_multi<true, true>), dim3(g1), dim3(64), 0, st, c->dG, c->dT, b->dB, (const int*)multiList, multiBase, scratch);
            } else {
                hipLaunchKernelGGL((k_pair_chains<false>), dim3(lean), dim3(64), 0, st, c->dG, c->dT, b->dB, (const uint8_t*)B.pair_deferred, mode, counterIdx, multiList, multiBase);
                hipLaunchKernelGGL((k_pair_multi<false, false>), dim3(g0), dim3(64), 0, st, c->dG, c->dT, b->dB, (const int*)multiList, multiBase, scratch);
                hipLaunchKernelGGL((k_pair_multi<false, true>), dim3(g1), dim3(64), 0, st, c->dG, c->dT, b->dB, (const int*)multiList, multiBase, scratch);
            }
            return check_launch(c, "k_pair_chains / k_pair_multi");
        };
        int rc = 0;
        if(phase != 2) { rc = launch_pair(c->active, fused ? 1 : 0, 2); if(rc) return rc; }
        if(fused && phase != 1) {
            // second pass, behind the side-stream classes and the second stitch pass: the deferred pairs (work counter 37)
            rc = launch_pair(c->side, 2, 37); if(rc) return rc;
            HIP_TRY(c, hipEventRecord(b->evSide[6], c->side));
            HIP_TRY(c, hipEventRecord(b->evDone, c->side));
            HIP_TRY(c, hipEventRecord(c->evSideTail, c->side)); c->sideTailValid = true;
            b->side_inflight = true; b->side_pending = false;
        }
    }
    if(phase == 2) return HLALA_OK;
    HIP_TRY(c, hipEventRecord(b->ev[5], c->active));
    b->staged |= 4;
    return mark_main(c, b);
}

// The tail pool (round 6).  The broad, large and in-memory DP classes hold 0.07 % of a batch's DP calls and took 100 ms of side stream per batch: their cost is the
// latency of their slowest calls, paid per launch, while their blocks hold most of every CU's LDS beside the next batch's kernels.  With hlala_set_tail_pool(ctx, k)
// a fused alignment runs its main-stream part and its wide class as before and leaves those three classes PENDING; the k-th pending batch (or hlala_flush, or any
// call that reads or re-runs a pending batch) launches each of them once over all pending batches (k_dp: DpPoolArgs), then the second stitch / pairing pass of every
// batch.  Results do not depend on k (tests/test_graph_m.py).
static int flush_tail(hlala_ctx* c)
{
    if(c->tail.empty()) return HLALA_OK;
    DEV_GUARD(c);
    std::vector<hlala_batch*> pool; pool.swap(c->tail);
    for(hlala_batch* b : pool) b->tail_pooled = false;           // (whatever happens below, nobody waits for this flush again)
    DpPoolArgs a{}; a.n = (int)pool.size();
    for(int i = 0; i < a.n; i++) { hlala_batch* b = pool[i]; a.seed[i] = c->params.rng_seed + 2u * b->first_chain; a.B[i] = b->dB; a.items[i] = (const DpItem*)b->B.dp_items; a.bases[i] = (const uint8_t*)b->B.read_bases; }
    const hipStream_t ws = c->side;
    for(int tier = DP_POOL_TIER; tier <= DP_LAST_TIER; tier++) {
        for(hlala_batch* b : pool) HIP_TRY(c, hipEventRecord(b->evC[tier][0], ws));
        hlala_batch* b0 = pool[0];
        switch(tier) {
        case 4: hipLaunchKernelGGL((k_dp<DpBroad, 4>), dim3(c->broad_grid), dim3(DpBroad::THREADS), 0, ws, c->dG, b0->dB, (DpItem*)b0->B.dp_items, c->large_slabs, c->large_slab_bytes, a.seed[0], c->G.nrec_out, c->G.nrec_in, b0->B.read_bases, a); break;
        case 5: hipLaunchKernelGGL((k_dp<DpLarge, 5>), dim3(c->retry_grid), dim3(DpLarge::THREADS), 0, ws, c->dG, b0->dB, (DpItem*)b0->B.dp_items, c->large_slabs + c->large_slab_bytes * (size_t)c->broad_grid, c->large_slab_bytes, a.seed[0], c->G.nrec_out, c->G.nrec_in, b0->B.read_bases, a); break;
        default: hipLaunchKernelGGL((k_dp<DpHuge, 6>), dim3(c->huge_grid), dim3(DpHuge::THREADS), 0, ws, c->dG, b0->dB, (DpItem*)b0->B.dp_items, c->huge_slabs, c->huge_slab_bytes, a.seed[0], c->G.nrec_out, c->G.nrec_in, b0->B.read_bases, a); break;
        }
        int rc_ = check_launch(c, "k_dp (pooled)"); if(rc_) return rc_;
        for(hlala_batch* b : pool) HIP_TRY(c, hipEventRecord(b->evC[tier][1], ws));
    }
    for(hlala_batch* b : pool) {
        int rc = extend_impl(c, b, true, 4); if(rc) return rc;      // second stitch pass over the deferred pairs
        rc = pair_impl(c, b, 2); if(rc) return rc;                   // second pairing pass; records evDone
    }
    return HLALA_OK;
}

int hlala_set_tail_pool(hlala_ctx* c, int k)
{
    if(!c || k < 1 || k > DP_POOL_MAX) { if(c) c->err = "hlala_set_tail_pool: k must be 1 .. " + std::to_string(DP_POOL_MAX); return HLALA_E_ARG; }
    if(k < (int)c->tail.size() || k == 1) { int rf = flush_tail(c); if(rf) return rf; }
    c->tail_pool_k = k;
    return HLALA_OK;
}
int hlala_flush(hlala_ctx* c)
{
    if(!c) return HLALA_E_ARG;
    return flush_tail(c);
}

int hlala_align_batch(hlala_ctx* c, hlala_batch* b)
{
    int rc = hlala_project_chains(c, b); if(rc) return rc;
    if(c->tail_pool_k > 1 && DP_POOL_TIER > DP_SIDE_TIER) {
        rc = extend_impl(c, b, true, 3); if(rc) return rc;
        if(!b->side_pending) return hlala_pair_chains(c, b);          // not a fused alignment (unpaired, from seeds, empty): nothing to pool
        rc = pair_impl(c, b, 1); if(rc) return rc;
        c->tail.push_back(b); b->tail_pooled = true;
        if((int)c->tail.size() >= c->tail_pool_k) return flush_tail(c);
        return HLALA_OK;
    }
    if(!c->side_after_pair) {
        rc = extend_impl(c, b, true); if(rc) return rc;
        return hlala_pair_chains(c, b);
    }
    // the main stream's part of the batch first -- DP classes up to the 64-lane one, stitch, pairing --, then the side stream's classes and its second stitch / pairing pass
    // are queued behind it (extend_impl): the main stream's short kernels do not share the CUs with the wide class
    rc = extend_impl(c, b, true, 1); if(rc) return rc;
    rc = pair_impl(c, b, 1); if(rc) return rc;
    rc = extend_impl(c, b, true, 2); if(rc) return rc;
    return pair_impl(c, b, 2);
}

int hlala_batch_get_chains(hlala_ctx* c, hlala_batch* b, int stage, hlala_chains_out* o)
{
    DEV_GUARD(c);
    ReaderScope rscope_(c, b); if(rscope_.rc) return rscope_.rc;
    if(!c || !b || !o) return HLALA_E_ARG;
    DevBatch& B = b->B;
    size_t nc = (size_t)B.n_chains, cs = nc * (size_t)B.stride;
    int rc = 0;
    // the column arrays hold a row per chain that passed the filters (batch.h: chain_row); the caller's arrays hold one per chain: rows are copied to their chains'
    // places on the host (chains without a row: zeros)
    std::vector<int> rowOf;
    if(B.chain_row && nc > 0) { rowOf.resize(nc); HIP_TRY(c, hipMemcpyAsync(rowOf.data(), B.chain_row, nc * sizeof(int), hipMemcpyDeviceToHost, c->active)); HIP_TRY(c, hipStreamSynchronize(c->active)); }
    auto dlRows = [&](auto* host, const auto* dev) -> int {
        typedef typename std::remove_pointer<decltype(host)>::type T;
        if(!host || !cs) return 0;
        if(rowOf.empty()) return dl(c, host, dev, cs);
        const size_t stride = (size_t)B.stride, nrows = (size_t)B.n_rows;
        std::vector<T> tmp(nrows * stride + 1);
        if(nrows) { HIP_TRY(c, hipMemcpyAsync(tmp.data(), dev, nrows * stride * sizeof(T), hipMemcpyDeviceToHost, c->active)); HIP_TRY(c, hipStreamSynchronize(c->active)); }
        for(size_t k = 0; k < nc; k++) {
            const int r = rowOf[k];
            if(r >= 0 && (size_t)r < nrows) memcpy(host + k * stride, tmp.data() + (size_t)r * stride, stride * sizeof(T)); else memset(host + k * stride, 0, stride * sizeof(T));
        }
        return 0;
    };
    if(stage == 0) {
        if(!(b->staged & 1)) { c->err = "seed chains not computed"; return HLALA_E_STATE; }
        if((rc = dl(c, o->status, B.seed_status, nc))) return rc; if((rc = dl(c, o->n_cols, B.seed_ncols, nc))) return rc;
        if((rc = dl(c, o->seq_begin, B.seed_begin, nc))) return rc; if((rc = dl(c, o->seq_end, B.seed_end, nc))) return rc;
        if((rc = dl(c, o->removed_cols, B.seed_removed, nc))) return rc;
        if((rc = dlRows(o->col_level, (const int*)B.seed_level))) return rc; if((rc = dlRows(o->col_edge, (const int*)B.seed_edge))) return rc;
        if((rc = dlRows(o->col_gchar, (const uint8_t*)B.seed_g))) return rc; if((rc = dlRows(o->col_schar, (const uint8_t*)B.seed_s))) return rc;
        HIP_TRY(c, hipStreamSynchronize(c->active));
        if(o->col_fromseed) memset(o->col_fromseed, 1, cs);
        if(o->ll) memset(o->ll, 0, nc * 8);
        if(o->dp_iters) memset(o->dp_iters, 0, nc * 8);
        if(o->dp_score) memset(o->dp_score, 0, nc * 8);
    } else if(stage == 1) {
        if(!(b->staged & 2)) { c->err = "extended chains not computed"; return HLALA_E_STATE; }
        if((rc = dl(c, o->status, B.ext_status, nc))) return rc; if((rc = dl(c, o->n_cols, B.ext_ncols, nc))) return rc;
        if((rc = dl(c, o->seq_begin, B.ext_begin, nc))) return rc; if((rc = dl(c, o->seq_end, B.ext_end, nc))) return rc;
        if((rc = dl(c, o->removed_cols, B.seed_removed, nc))) return rc; if((rc = dl(c, o->ll, B.ext_ll, nc))) return rc;
        if((rc = dl(c, o->dp_iters, B.dp_iters, 2 * nc))) return rc; if((rc = dl(c, o->dp_score, B.dp_score, 2 * nc))) return rc;
        if((rc = dlRows(o->col_level, (const int*)B.ext_level))) return rc; if((rc = dlRows(o->col_edge, (const int*)B.ext_edge))) return rc;
        if((rc = dlRows(o->col_gchar, (const uint8_t*)B.ext_g))) return rc; if((rc = dlRows(o->col_schar, (const uint8_t*)B.ext_s))) return rc;
        if((rc = dlRows(o->col_fromseed, (const uint8_t*)B.ext_fromseed))) return rc;
        HIP_TRY(c, hipStreamSynchronize(c->active));
    } else { c->err = "stage must be 0 or 1"; return HLALA_E_ARG; }
    return HLALA_OK;
}

int hlala_batch_get_pairs(hlala_ctx* c, hlala_batch* b, hlala_pairs_out* o)
{
    DEV_GUARD(c);
    ReaderScope rscope_(c, b); if(rscope_.rc) return rscope_.rc;
    if(!c || !b || !o) return HLALA_E_ARG;
    if(!(b->staged & 4)) { c->err = "pairs not computed"; return HLALA_E_STATE; }
    DevBatch& B = b->B;
    size_t np = (size_t)B.n_pairs, nr = (size_t)B.n_reads, stride = (size_t)B.stride;
    int rc = 0;
    if((rc = dl(c, o->best_chain, B.best_chain, nr))) return rc;
    if((rc = dl(c, o->pair_status, B.pair_status, np))) return rc; if((rc = dl(c, o->n_combinations, B.n_comb, np))) return rc;
    if((rc = dl(c, o->pair_ll, B.pair_ll, np))) return rc; if((rc = dl(c, o->pair_mapq, B.pair_mapq, np))) return rc;
    if((rc = dl(c, o->mate_mapq, B.mate_mapq, nr))) return rc; if((rc = dl(c, o->strands_valid, B.strands_valid, np))) return rc;
    if((rc = dl(c, o->col_mapq, B.sel_mapq, nr * stride))) return rc;
    HIP_TRY(c, hipStreamSynchronize(c->active));
    // columns of the selected chains: gathered on the device into read-major staging rows, one bulk copy per array and chunk
    // (columns beyond n_cols come back as zero)
    const bool wantCols = o->n_cols || o->col_level || o->col_edge || o->col_gchar || o->col_schar || o->col_fromseed;
    if(wantCols && nr > 0) {
        const size_t CH = nr < 65536 ? nr : 65536;
        std::vector<void*> tmp;
        auto done = [&](int r_) { for(void* p : tmp) pool_release(c, p); return r_; };
        int* dN = nullptr; int* dL = nullptr; int* dE = nullptr; uint8_t* dG = nullptr; uint8_t* dS = nullptr; uint8_t* dF = nullptr;
        if((rc = dev_alloc(c, tmp, CH, &dN, false)) || (rc = dev_alloc(c, tmp, CH * stride, &dL, false)) || (rc = dev_alloc(c, tmp, CH * stride, &dE, false)) ||
           (rc = dev_alloc(c, tmp, CH * stride, &dG, false)) || (rc = dev_alloc(c, tmp, CH * stride, &dS, false)) || (rc = dev_alloc(c, tmp, CH * stride, &dF, false))) return done(rc);
        for(size_t r0 = 0; r0 < nr; r0 += CH) {
            const size_t rows = nr - r0 < CH ? nr - r0 : CH;
            hipLaunchKernelGGL(k_gather_selected, dim3((unsigned)rows), dim3(128), 0, c->active, b->dB, (int)r0, (int)rows, dN, dL, dE, dG, dS, dF);
            if((rc = check_launch(c, "k_gather_selected"))) return done(rc);
            const size_t cols = rows * stride, o0 = r0 * stride;
            hipError_t e = hipSuccess;
            if(o->n_cols && e == hipSuccess) e = hipMemcpyAsync(o->n_cols + r0, dN, rows * 4, hipMemcpyDeviceToHost, c->active);
            if(o->col_level && e == hipSuccess) e = hipMemcpyAsync(o->col_level + o0, dL, cols * 4, hipMemcpyDeviceToHost, c->active);
            if(o->col_edge && e == hipSuccess) e = hipMemcpyAsync(o->col_edge + o0, dE, cols * 4, hipMemcpyDeviceToHost, c->active);
            if(o->col_gchar && e == hipSuccess) e = hipMemcpyAsync(o->col_gchar + o0, dG, cols, hipMemcpyDeviceToHost, c->active);
            if(o->col_schar && e == hipSuccess) e = hipMemcpyAsync(o->col_schar + o0, dS, cols, hipMemcpyDeviceToHost, c->active);
            if(o->col_fromseed && e == hipSuccess) e = hipMemcpyAsync(o->col_fromseed + o0, dF, cols, hipMemcpyDeviceToHost, c->active);
            if(e == hipSuccess) e = hipStreamSynchronize(c->active);
            if(e != hipSuccess) { c->err = std::string("hlala_batch_get_pairs: ") + hipGetErrorString(e); return done(HLALA_E_DEVICE); }
        }
        done(0);
    }
    return HLALA_OK;
}

int hlala_batch_get_pairs_packed(hlala_ctx* c, hlala_batch* b, hlala_pairs_packed_out* o)
{
    DEV_GUARD(c);
    ReaderScope rscope_(c, b); if(rscope_.rc) return rscope_.rc;
    if(!c || !b || !o || !o->col_off) return HLALA_E_ARG;
    if(!(b->staged & 4)) { c->err = "pairs not computed"; return HLALA_E_STATE; }
    DevBatch& B = b->B;
    const size_t nr = (size_t)B.n_reads;
    o->n_cols_total = 0;
    if(nr == 0) { o->col_off[0] = 0; return HLALA_OK; }
    std::vector<void*> tmp;
    auto done = [&](int r_) { for(void* p : tmp) pool_release(c, p); return r_; };
    int rc = 0; long long *dN = nullptr, *dOff = nullptr; char* dCub = nullptr;
    if((rc = dev_alloc(c, tmp, nr + 1, &dN)) || (rc = dev_alloc(c, tmp, nr + 1, &dOff))) return done(rc);
    hipStream_t st = c->active;
    hipLaunchKernelGGL(k_selected_ncols, dim3((unsigned)((nr + 1 + 255) / 256)), dim3(256), 0, st, b->dB, (int)nr, dN);
    if((rc = check_launch(c, "k_selected_ncols"))) return done(rc);
    size_t cubBytes = 0;
    HIP_TRY_F(c, hipcub::DeviceScan::ExclusiveSum(nullptr, cubBytes, dN, dOff, (int)(nr + 1), st), done);
    if((rc = dev_alloc(c, tmp, cubBytes ? cubBytes : 1, &dCub))) return done(rc);
    HIP_TRY_F(c, hipcub::DeviceScan::ExclusiveSum(dCub, cubBytes, dN, dOff, (int)(nr + 1), st), done);
    long long total = 0;
    HIP_TRY_F(c, hipMemcpyAsync(&total, dOff + nr, sizeof(total), hipMemcpyDeviceToHost, st), done);
    HIP_TRY_F(c, hipStreamSynchronize(st), done);
    o->n_cols_total = total;
    { std::vector<long long> ho(nr + 1); if((rc = dl(c, ho.data(), dOff, nr + 1))) return done(rc); HIP_TRY_F(c, hipStreamSynchronize(st), done); for(size_t i = 0; i <= nr; i++) o->col_off[i] = ho[i]; }
    if(total > o->cap_cols) { c->err = "hlala_batch_get_pairs_packed: cap_cols too small (n_cols_total holds the need)"; return done(HLALA_E_CAPACITY); }
    if(total == 0) return done(HLALA_OK);
    int *dL = nullptr, *dE = nullptr; uint8_t *dG = nullptr, *dS = nullptr, *dF = nullptr, *dQ = nullptr;
    const size_t T = (size_t)total;
    if((o->col_level && (rc = dev_alloc(c, tmp, T, &dL))) || (o->col_edge && (rc = dev_alloc(c, tmp, T, &dE))) || (o->col_gchar && (rc = dev_alloc(c, tmp, T, &dG))) ||
       (o->col_schar && (rc = dev_alloc(c, tmp, T, &dS))) || (o->col_fromseed && (rc = dev_alloc(c, tmp, T, &dF))) || (o->col_mapq && (rc = dev_alloc(c, tmp, T, &dQ)))) return done(rc);
    const unsigned grid = (unsigned)std::min<size_t>(nr, (size_t)c->stitch_grid * 4);
    hipLaunchKernelGGL(k_gather_packed, dim3(grid), dim3(64), 0, st, b->dB, (int)nr, (const long long*)dOff, dL, dE, dG, dS, dF, dQ);
    if((rc = check_launch(c, "k_gather_packed"))) return done(rc);
    if((rc = dl(c, o->col_level, dL, T)) || (rc = dl(c, o->col_edge, dE, T)) || (rc = dl(c, o->col_gchar, dG, T)) || (rc = dl(c, o->col_schar, dS, T)) ||
       (rc = dl(c, o->col_fromseed, dF, T)) || (rc = dl(c, o->col_mapq, dQ, T))) return done(rc);
    HIP_TRY_F(c, hipStreamSynchronize(st), done);
    return done(HLALA_OK);
}

int hlala_estimate_insert_size(hlala_ctx* c, const hlala_batch_in* in, hlala_insert_size_out* out)
{
    DEV_GUARD(c);
    if(!c || !in || !out) return HLALA_E_ARG;
    memset(out, 0, sizeof(*out));
    const int np = in->n_pairs, nr = 2 * np;
    if(np <= 0) { c->err = "hlala_estimate_insert_size: no pairs"; return HLALA_E_ARG; }
    // the batch of primaries: one chain per read (chain indices of `in` are those of the caller's numbering: the window convention of hlala_batch_in)
    std::vector<int64_t> chain_off(nr + 1), cigar_off(nr + 1, 0); std::vector<int32_t> read_primary(nr), contig(nr), pos(nr), offset(nr), as(nr); std::vector<uint8_t> rev(nr); std::vector<uint32_t> cigar;
    for(int r = 0; r < nr; r++) {
        const int64_t ch = in->read_primary[r];
        if(ch < in->chain_off[r] || ch >= in->chain_off[r + 1]) { c->err = "read_primary outside the read's chains"; return HLALA_E_ARG; }
        chain_off[r] = r; read_primary[r] = r; contig[r] = in->chain_contig[ch]; pos[r] = in->chain_pos[ch]; offset[r] = in->chain_offset[ch]; as[r] = in->chain_as[ch]; rev[r] = in->chain_reverse[ch];
        cigar.insert(cigar.end(), in->cigar + in->cigar_off[ch], in->cigar + in->cigar_off[ch + 1]); cigar_off[r + 1] = (int64_t)cigar.size();
    }
    chain_off[nr] = nr;
    hlala_batch_in pb = *in;
    pb.chain_off = chain_off.data(); pb.read_primary = read_primary.data(); pb.n_chains = nr; pb.chain_contig = contig.data(); pb.chain_pos = pos.data();
    pb.chain_offset = offset.data(); pb.chain_as = as.data(); pb.chain_reverse = rev.data(); pb.cigar_off = cigar_off.data(); pb.cigar = cigar.data();
    hlala_batch* b = nullptr;
    int rc = hlala_batch_create(c, &pb, &b); if(rc) return rc;
    auto done = [&](int r_) { hlala_batch_destroy(b); return r_; };
    if((rc = hlala_project_chains(c, b)) || (rc = hlala_extend_chains(c, b))) return done(rc);
    std::vector<void*> tmp; int *dN = nullptr, *dD = nullptr;
    auto done2 = [&](int r_) { for(void* p : tmp) pool_release(c, p); return done(r_); };
    if((rc = dev_alloc(c, tmp, (size_t)np, &dN)) || (rc = dev_alloc(c, tmp, (size_t)np * PAIR_MAXDIST, &dD))) return done2(rc);
    hipLaunchKernelGGL(k_pair_distances, dim3((unsigned)((np + 127) / 128)), dim3(128), 0, c->active, c->dG, b->dB, dN, dD);
    if((rc = check_launch(c, "k_pair_distances"))) return done2(rc);
    std::vector<int> hN((size_t)np), hD((size_t)np * PAIR_MAXDIST);
    if((rc = dl(c, hN.data(), dN, (size_t)np)) || (rc = dl(c, hD.data(), dD, (size_t)np * PAIR_MAXDIST))) return done2(rc);
    hipError_t e = hipStreamSynchronize(c->active);
    if(e != hipSuccess) { c->err = hipGetErrorString(e); return done2(HLALA_E_DEVICE); }
    // histogram in pair order (processBAM.cpp:1135-1146), then calculateInsertSizeFromHistogram (:991-1069)
    std::map<int, double> IS_combined_counts;
    for(int p = 0; p < np; p++) {
        if(hN[p] == -1) { out->n_skipped++; continue; }                       // flagged chain: the reference would have asserted
        out->n_used++;
        if(hN[p] == -2) { out->n_skipped++; continue; }                       // strands not valid, :1149
        if(hN[p] > PAIR_MAXDIST) { c->err = "more distinct insert-size distances for one pair than this build holds"; return done2(HLALA_E_CAPACITY); }
        std::set<int> dist(hD.begin() + (size_t)p * PAIR_MAXDIST, hD.begin() + (size_t)p * PAIR_MAXDIST + hN[p]);
        for(int IS : dist) { if(IS_combined_counts.count(IS) == 0) IS_combined_counts[IS] = 0; IS_combined_counts[IS] += 1.0 / (double)dist.size(); }
    }
    double total = 0; for(auto& kv : IS_combined_counts) total += kv.second;
    double cum = 0, med = 0, p20 = 0, p80 = 0; bool sm = false, s2 = false, s8 = false;
    for(auto& kv : IS_combined_counts) {
        cum += kv.second;
        if(!sm && cum >= total * 0.5) { med = kv.first; sm = true; }
        if(!s2 && cum >= total * 0.2) { p20 = kv.first; s2 = true; }
        if(!s8 && cum >= total * 0.8) { p80 = kv.first; s8 = true; }
    }
    if(!(sm && s2 && s8)) { c->err = "hlala_estimate_insert_size: no pair with valid strands and a common underlying sequence"; return done2(HLALA_E_STATE); }
    const double d20 = std::fabs(med - p20), d80 = std::fabs(med - p80);
    out->mean = med; out->sd = d20 > d80 ? d20 : d80; out->total_weight = total;
    return done2(HLALA_OK);
}

int hlala_set_insert_size(hlala_ctx* c, double insert_mean, double insert_sd)
{
    DEV_GUARD(c);
    if(!c) return HLALA_E_ARG;
    HIP_TRY(c, hipStreamSynchronize(c->stream)); HIP_TRY(c, hipStreamSynchronize(c->side));
    const double m0 = c->params.insert_mean, s0 = c->params.insert_sd;
    double* oldLog = c->d_islog; DevTables* oldT = c->dT;
    c->params.insert_mean = insert_mean; c->params.insert_sd = insert_sd;
    // (the kernels read the tables through c->dT, which batches do not cache: the new tables go to the same device address)
    std::vector<void*> keep; keep.swap(c->allocs);
    int rc = build_tables(c);
    std::vector<void*> made; made.swap(c->allocs); c->allocs.swap(keep);
    if(rc) { for(void* p : made) if(p) pool_release(c, p); c->params.insert_mean = m0; c->params.insert_sd = s0; c->d_islog = oldLog; c->dT = oldT; return rc; }
    // copy the new table struct over the old one so that every holder of the old pointer sees it; the log-pdf array it points to stays alive in allocs
    hipError_t e = hipMemcpyAsync(oldT, c->dT, sizeof(DevTables), hipMemcpyDeviceToDevice, c->stream);
    if(e == hipSuccess) e = hipStreamSynchronize(c->stream);
    for(void* p : made) { if(p == (void*)c->dT) pool_release(c, p); else c->allocs.push_back(p); }
    c->dT = oldT;
    if(e != hipSuccess) { c->err = std::string("hlala_set_insert_size: ") + hipGetErrorString(e); return HLALA_E_DEVICE; }
    return HLALA_OK;
}

int hlala_set_gene_intervals(hlala_ctx* c, int32_t n, const int32_t* first_level, const int32_t* last_level)
{
    DEV_GUARD(c);
    if(!c || n < 0 || (n > 0 && (!first_level || !last_level))) return HLALA_E_ARG;
    for(int i = 0; i < n; i++) if(first_level[i] < 0 || last_level[i] < first_level[i]) { c->err = "gene interval with first > last or negative level"; return HLALA_E_ARG; }
    c->n_genes = 0;
    if(n > 0) {
        int rc = dev_upload(c, c->allocs, first_level, (size_t)n, &c->d_gene_first); if(rc) return rc;
        rc = dev_upload(c, c->allocs, last_level, (size_t)n, &c->d_gene_last); if(rc) return rc;
        HIP_TRY(c, hipStreamSynchronize(c->active));
    }
    c->n_genes = n;
    return HLALA_OK;
}

int hlala_postprocess_pairs(hlala_ctx* c, hlala_batch* b, uint8_t* include_in_hla)
{
    DEV_GUARD(c);
    ReaderScope rscope_(c, b); if(rscope_.rc) return rscope_.rc;
    if(!c || !b) return HLALA_E_ARG;
    if(!(b->staged & 4)) { c->err = "hlala_postprocess_pairs before hlala_pair_chains"; return HLALA_E_STATE; }
    DevBatch& B = b->B;
    if(!c->d_cov) {
        c->n_cov = c->F.L > 1 ? c->F.L - 1 : 1;
        int rc = dev_alloc(c, c->allocs, (size_t)c->n_cov, &c->d_cov, true); if(rc) return rc;
    }
    if(B.n_pairs <= 0) return HLALA_OK;
    uint8_t* dInc = nullptr; std::vector<void*> tmp;
    auto done = [&](int r_) { for(void* p : tmp) pool_release(c, p); return r_; };
    if(include_in_hla) { int rc = dev_alloc(c, tmp, (size_t)B.n_pairs, &dInc, false); if(rc) return done(rc); }
    hipError_t e = hipMemsetAsync(B.work_counter + 11, 0, sizeof(int), c->active);
    if(e != hipSuccess) { c->err = hipGetErrorString(e); return done(HLALA_E_DEVICE); }
    int grid = B.n_pairs < c->stitch_grid ? B.n_pairs : c->stitch_grid;
    hipLaunchKernelGGL(k_post_pairs, dim3(grid), dim3(64), 0, c->active, b->dB, c->d_cov, c->n_cov, c->d_gene_first, c->d_gene_last, c->n_genes, dInc);
    int rc = check_launch(c, "k_post_pairs"); if(rc) return done(rc);
    if(include_in_hla) e = hipMemcpyAsync(include_in_hla, dInc, (size_t)B.n_pairs, hipMemcpyDeviceToHost, c->active);
    if(e == hipSuccess) e = hipStreamSynchronize(c->active);
    if(e != hipSuccess) { c->err = std::string("hlala_postprocess_pairs: ") + hipGetErrorString(e); return done(HLALA_E_DEVICE); }
    return done(HLALA_OK);
}

int hlala_get_coverage(hlala_ctx* c, int32_t* bases_per_level, int reset)
{
    DEV_GUARD(c);
    ReaderScope rscope_(c, nullptr);
    if(!c || !bases_per_level) return HLALA_E_ARG;
    const int n = c->F.L > 1 ? c->F.L - 1 : 1;
    if(!c->d_cov) { memset(bases_per_level, 0, (size_t)n * 4); return HLALA_OK; }
    HIP_TRY(c, hipMemcpyAsync(bases_per_level, c->d_cov, (size_t)n * 4, hipMemcpyDeviceToHost, c->active));
    if(reset) HIP_TRY(c, hipMemsetAsync(c->d_cov, 0, (size_t)n * 4, c->active));
    HIP_TRY(c, hipStreamSynchronize(c->active));
    return HLALA_OK;
}

int hlala_batch_export_pair_records(hlala_ctx* c, hlala_batch* b, double* device_out)
{
    DEV_GUARD(c);
    // a reader like the getters: on the reader stream, behind THIS batch's work on the main and the side stream -- not behind the alignment of the next batch the
    // caller has queued on the main stream since (it ran there until round 4: with two batches in flight the records of batch i waited for batch i+1) -- and
    // synchronised on return, so that a collective the caller issues next on any stream of its own finds the records in place
    ReaderScope rscope_(c, b); if(rscope_.rc) return rscope_.rc;
    if(!c || !b || !device_out) return HLALA_E_ARG;
    if(!(b->staged & 4)) { c->err = "pairs not computed"; return HLALA_E_STATE; }
    if(b->B.n_pairs > 0) {
        hipLaunchKernelGGL(k_export_pairs, dim3((b->B.n_pairs + 255) / 256), dim3(256), 0, c->active, b->dB, device_out);
        int rc = check_launch(c, "k_export_pairs"); if(rc) return rc;
        HIP_TRY(c, hipStreamSynchronize(c->active));
    }
    return HLALA_OK;
}

// ---- several GPUs in ONE process (the host program's --devices: one context per GPU): the exchange steps of the path over RCCL.
// The reference merges its per-thread results on the host (mapper/processBAM.cpp:1866-1887: the per-thread alignment vectors are appended, the per-level read
// counters added up); with one context per GPU the same two steps are a gather of the per-pair records and a sum-reduce of the coverage counters to the first
// context's device.  RCCL is looked up at run time (dlopen: a one-GPU run never loads it) and used when the communicator's contexts sit on different devices;
// contexts that share a device (tests on a one-GPU box) or a communicator of one take device-to-device copies on the same arrays.
#include <rccl/rccl.h>
#include <dlfcn.h>
namespace {
struct RcclApi {
    void* lib = nullptr; bool tried = false; std::string err;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr; ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Reduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    bool load()
    {
        if(tried) return lib != nullptr;
        tried = true;
        for(const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) { lib = dlopen(name, RTLD_NOW | RTLD_LOCAL); if(lib) break; }
        if(!lib) { err = std::string("librccl.so not found: ") + dlerror(); return false; }
#define RSYM(field, sym) do { *(void**)(&field) = dlsym(lib, sym); if(!field) { err = std::string("librccl.so lacks ") + sym; dlclose(lib); lib = nullptr; return false; } } while(0)
        RSYM(CommInitAll, "ncclCommInitAll"); RSYM(CommDestroy, "ncclCommDestroy"); RSYM(GroupStart, "ncclGroupStart"); RSYM(GroupEnd, "ncclGroupEnd");
        RSYM(Send, "ncclSend"); RSYM(Recv, "ncclRecv"); RSYM(Reduce, "ncclReduce"); RSYM(GetErrorString, "ncclGetErrorString");
#undef RSYM
        return true;
    }
};
RcclApi g_rccl; std::mutex g_rccl_mu;
}
struct hlala_comm {
    std::vector<hlala_ctx*> ctxs;
    std::vector<ncclComm_t> comms;            // empty: device-to-device copies (one context, or contexts that share a device)
    std::vector<hipStream_t> streams;         // one per context, for the collectives
    std::vector<double*> send; std::vector<size_t> sendCap;      // per context: its batch's records
    double* recv = nullptr; size_t recvCap = 0;                  // on the first context's device: every context's records, in context order
    int* covAcc = nullptr; size_t covN = 0;                      // ... and the summed coverage counters
    std::string err;
};
static thread_local std::string g_comm_error;
const char* hlala_comm_last_error(const hlala_comm* m) { return m ? m->err.c_str() : g_comm_error.c_str(); }
int hlala_comm_uses_rccl(const hlala_comm* m) { return m && !m->comms.empty() ? 1 : 0; }

void hlala_comm_destroy(hlala_comm* m)
{
    if(!m) return;
    for(size_t i = 0; i < m->ctxs.size(); i++) {
        DevGuard g(m->ctxs[i]->device);
        if(i < m->streams.size() && m->streams[i]) { (void)hipStreamSynchronize(m->streams[i]); (void)hipStreamDestroy(m->streams[i]); }
        if(i < m->send.size() && m->send[i]) (void)hipFree(m->send[i]);
        if(i == 0) { if(m->recv) (void)hipFree(m->recv); if(m->covAcc) (void)hipFree(m->covAcc); }
    }
    for(ncclComm_t q : m->comms) if(q) (void)g_rccl.CommDestroy(q);
    delete m;
}

int hlala_comm_create(hlala_ctx* const* ctxs, int n, hlala_comm** out)
{
    if(out) *out = nullptr;
    if(!ctxs || n < 1 || !out) { g_comm_error = "hlala_comm_create: no contexts"; return HLALA_E_ARG; }
    for(int i = 0; i < n; i++) if(!ctxs[i]) { g_comm_error = "hlala_comm_create: null context"; return HLALA_E_ARG; }
    hlala_comm* m = new hlala_comm;
    m->ctxs.assign(ctxs, ctxs + n); m->streams.assign((size_t)n, nullptr); m->send.assign((size_t)n, nullptr); m->sendCap.assign((size_t)n, 0);
    auto fail = [&](int rc, const std::string& e) { g_comm_error = e; hlala_comm_destroy(m); return rc; };
    for(int i = 0; i < n; i++) { DevGuard g(ctxs[i]->device); if(hipStreamCreateWithFlags(&m->streams[(size_t)i], hipStreamNonBlocking) != hipSuccess) return fail(HLALA_E_DEVICE, "hlala_comm_create: hipStreamCreate failed"); }
    std::set<int> distinct; for(int i = 0; i < n; i++) distinct.insert(ctxs[i]->device);
    const char* force = getenv("HLALA_COMM_RCCL");          // 1: RCCL even for a communicator of one (tests); 0: never
    const bool want = force ? atoi(force) != 0 : n > 1;
    if(want && (int)distinct.size() == n) {
        std::lock_guard<std::mutex> g(g_rccl_mu);
        if(!g_rccl.load()) return fail(HLALA_E_DEVICE, "hlala_comm_create: " + g_rccl.err);
        std::vector<int> devs((size_t)n); for(int i = 0; i < n; i++) devs[(size_t)i] = ctxs[i]->device;
        m->comms.assign((size_t)n, nullptr);
        const ncclResult_t r = g_rccl.CommInitAll(m->comms.data(), n, devs.data());
        if(r != ncclSuccess) { m->comms.clear(); return fail(HLALA_E_DEVICE, std::string("ncclCommInitAll: ") + g_rccl.GetErrorString(r)); }
    }
    *out = m;
    return HLALA_OK;
}

static int comm_buf(hlala_comm* m, int device, double** p, size_t* cap, size_t need)
{
    if(*cap >= need && *p) return HLALA_OK;
    DevGuard g(device);
    if(*p) (void)hipFree(*p);
    *p = nullptr; *cap = 0;
    if(device_malloc_retry(device, nullptr, (void**)p, (need ? need : 1) * sizeof(double)) != hipSuccess) { m->err = "hipMalloc (gather buffers) failed"; return HLALA_E_DEVICE; }
    *cap = need;
    return HLALA_OK;
}

// The per-pair records (8 doubles per pair: hlala_batch_export_pair_records) of one batch per context -- batches[i] belongs to context i of the communicator, NULL:
// that context has none this round -- gathered to the first context's device and copied to host_out in context order; counts_out[i] = pairs of context i.
// The counts are known on the host (the batches live in this process: the "counts first" step of the multi-process protocol in hla-la_amd/dist.py needs no
// collective here); the payload is ONE grouped exchange: every context sends its records, the first one posts a receive per peer at that peer's offset.
int hlala_gather_pair_records(hlala_comm* m, hlala_batch* const* batches, double* host_out, int64_t host_capacity_pairs, int64_t* counts_out)
{
    if(!m || !batches || !host_out) return HLALA_E_ARG;
    const int n = (int)m->ctxs.size();
    std::vector<size_t> cnt((size_t)n, 0), off((size_t)n + 1, 0);
    for(int i = 0; i < n; i++) { if(batches[i]) { if(batches[i]->ctx != m->ctxs[(size_t)i]) { m->err = "hlala_gather_pair_records: batch " + std::to_string(i) + " does not belong to context " + std::to_string(i); return HLALA_E_ARG; } cnt[(size_t)i] = (size_t)batches[i]->B.n_pairs; } off[(size_t)i + 1] = off[(size_t)i] + cnt[(size_t)i]; if(counts_out) counts_out[i] = (int64_t)cnt[(size_t)i]; }
    const size_t total = off[(size_t)n];
    if((int64_t)total > host_capacity_pairs) { m->err = "hlala_gather_pair_records: " + std::to_string(total) + " pairs, room for " + std::to_string(host_capacity_pairs); return HLALA_E_CAPACITY; }
    if(total == 0) return HLALA_OK;
    int rc = comm_buf(m, m->ctxs[0]->device, &m->recv, &m->recvCap, 8 * total); if(rc) return rc;
    const bool rccl = !m->comms.empty();
    // every context's records into its send buffer (the first context's straight into its place in the receive buffer when no RCCL exchange follows)
    for(int i = 0; i < n; i++) {
        if(!cnt[(size_t)i]) continue;
        double* dst = nullptr;
        if(!rccl && m->ctxs[(size_t)i]->device == m->ctxs[0]->device) dst = m->recv + 8 * off[(size_t)i];
        else { rc = comm_buf(m, m->ctxs[(size_t)i]->device, &m->send[(size_t)i], &m->sendCap[(size_t)i], 8 * cnt[(size_t)i]); if(rc) return rc; dst = m->send[(size_t)i]; }
        rc = hlala_batch_export_pair_records(m->ctxs[(size_t)i], batches[i], dst);      // (returns synchronised: the records are in place)
        if(rc) { m->err = std::string("hlala_batch_export_pair_records: ") + m->ctxs[(size_t)i]->err; return rc; }
    }
    if(rccl) {
        ncclResult_t r = g_rccl.GroupStart();
        for(int i = 0; i < n && r == ncclSuccess; i++) {
            if(!cnt[(size_t)i]) continue;
            r = g_rccl.Send(m->send[(size_t)i], 8 * cnt[(size_t)i], ncclDouble, 0, m->comms[(size_t)i], m->streams[(size_t)i]);
            if(r == ncclSuccess) r = g_rccl.Recv(m->recv + 8 * off[(size_t)i], 8 * cnt[(size_t)i], ncclDouble, i, m->comms[0], m->streams[0]);
        }
        const ncclResult_t re = g_rccl.GroupEnd();
        if(r == ncclSuccess) r = re;
        if(r != ncclSuccess) { m->err = std::string("RCCL gather: ") + g_rccl.GetErrorString(r); return HLALA_E_DEVICE; }
        for(int i = 1; i < n; i++) if(cnt[(size_t)i]) { DevGuard g(m->ctxs[(size_t)i]->device); if(hipStreamSynchronize(m->streams[(size_t)i]) != hipSuccess) { m->err = "RCCL gather: send stream failed"; return HLALA_E_DEVICE; } }
    } else {
        // contexts on other devices without RCCL (HLALA_COMM_RCCL=0): peer copies into the receive buffer
        for(int i = 0; i < n; i++) if(cnt[(size_t)i] && m->ctxs[(size_t)i]->device != m->ctxs[0]->device) {
            DevGuard g(m->ctxs[0]->device);
            if(hipMemcpyPeerAsync(m->recv + 8 * off[(size_t)i], m->ctxs[0]->device, m->send[(size_t)i], m->ctxs[(size_t)i]->device, 8 * cnt[(size_t)i] * sizeof(double), m->streams[0]) != hipSuccess) { m->err = "hipMemcpyPeerAsync failed"; return HLALA_E_DEVICE; }
        }
    }
    DevGuard g0(m->ctxs[0]->device);
    if(hipMemcpyAsync(host_out, m->recv, 8 * total * sizeof(double), hipMemcpyDeviceToHost, m->streams[0]) != hipSuccess || hipStreamSynchronize(m->streams[0]) != hipSuccess) { m->err = "gather: copy to the host failed"; return HLALA_E_DEVICE; }
    return HLALA_OK;
}

// bases_per_level of every context of the communicator added up on the first context's device (ncclReduce, sum of int32 -- integer sums do not depend on the
// order) and copied to bases_per_level; reset: the contexts' counters are cleared afterwards (hlala_get_coverage).  processBAM.cpp:1866-1887, :1902-1913.
int hlala_reduce_coverage(hlala_comm* m, int32_t* bases_per_level, int reset)
{
    if(!m || !bases_per_level) return HLALA_E_ARG;
    const int n = (int)m->ctxs.size();
    hlala_ctx* c0 = m->ctxs[0];
    const size_t nl = (size_t)(c0->F.L > 1 ? c0->F.L - 1 : 1);
    for(int i = 0; i < n; i++) if(m->ctxs[(size_t)i]->F.L != c0->F.L) { m->err = "hlala_reduce_coverage: the contexts hold different graphs"; return HLALA_E_ARG; }
    if(m->comms.empty() || n == 1) {
        // one context, or contexts sharing a device: the host adds the counters up
        std::vector<int32_t> one(nl);
        std::fill(bases_per_level, bases_per_level + nl, 0);
        for(int i = 0; i < n; i++) { int rc = hlala_get_coverage(m->ctxs[(size_t)i], one.data(), reset); if(rc) { m->err = m->ctxs[(size_t)i]->err; return rc; } for(size_t k = 0; k < nl; k++) bases_per_level[k] += one[k]; }
        return HLALA_OK;
    }
    { DevGuard g(c0->device); if(m->covN < nl) { if(m->covAcc) (void)hipFree(m->covAcc); m->covAcc = nullptr; if(device_malloc_retry(c0->device, nullptr, (void**)&m->covAcc, nl * sizeof(int)) != hipSuccess) { m->err = "hipMalloc (coverage) failed"; return HLALA_E_DEVICE; } m->covN = nl; } }
    // every context's counters exist and are final: its queued work is waited for first (post-processing adds to them on the reader stream)
    for(int i = 0; i < n; i++) { hlala_ctx* c = m->ctxs[(size_t)i]; DevGuard g(c->device);
        if(!c->d_cov) { c->n_cov = (int)nl; int rc = dev_alloc(c, c->allocs, nl, &c->d_cov, true); if(rc) { m->err = c->err; return rc; } }
        if(hipStreamSynchronize(c->rs) != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess) { m->err = "hlala_reduce_coverage: a context's stream failed"; return HLALA_E_DEVICE; } }
    ncclResult_t r = g_rccl.GroupStart();
    for(int i = 0; i < n && r == ncclSuccess; i++) r = g_rccl.Reduce(m->ctxs[(size_t)i]->d_cov, i == 0 ? (void*)m->covAcc : nullptr, nl, ncclInt32, ncclSum, 0, m->comms[(size_t)i], m->streams[(size_t)i]);
    const ncclResult_t re = g_rccl.GroupEnd();
    if(r == ncclSuccess) r = re;
    if(r != ncclSuccess) { m->err = std::string("RCCL reduce: ") + g_rccl.GetErrorString(r); return HLALA_E_DEVICE; }
    for(int i = 0; i < n; i++) { hlala_ctx* c = m->ctxs[(size_t)i]; DevGuard g(c->device);
        if(hipStreamSynchronize(m->streams[(size_t)i]) != hipSuccess) { m->err = "RCCL reduce: stream failed"; return HLALA_E_DEVICE; }
        if(reset && hipMemset(c->d_cov, 0, nl * sizeof(int)) != hipSuccess) { m->err = "hipMemset failed"; return HLALA_E_DEVICE; } }
    DevGuard g0(c0->device);
    if(hipMemcpy(bases_per_level, m->covAcc, nl * sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) { m->err = "coverage: copy to the host failed"; return HLALA_E_DEVICE; }
    return HLALA_OK;
}


int hlala_batch_get_stats(hlala_ctx* c, hlala_batch* b, hlala_batch_stats* out)
{
    DEV_GUARD(c);
    ReaderScope rscope_(c, b); if(rscope_.rc) return rscope_.rc;
    if(!c || !b || !out) return HLALA_E_ARG;
    memset(out, 0, sizeof(*out));
    if(!b->outputs_ready) return HLALA_OK;       // only uploaded so far: no stage has run, the counters do not exist yet (zeros, as before the outputs were allocated lazily)
    HIP_TRY(c, hipStreamSynchronize(c->active));
    u64 cnt[16];
    HIP_TRY(c, hipMemcpyAsync(cnt, b->B.counters, sizeof(cnt), hipMemcpyDeviceToHost, c->active)); HIP_TRY(c, hipStreamSynchronize(c->active));
    if((b->staged & 1) && !b->B.from_seeds) (void)hipEventElapsedTime(&out->ms_project, b->ev[0], b->ev[1]);
    if(b->staged & 2) { (void)hipEventElapsedTime(&out->ms_extend, b->ev[2], b->ev[3]); if(b->B.n_chains > 0) { (void)hipEventElapsedTime(&out->ms_extend_retry, b->ev[6], b->side_used ? b->ev[10] : b->ev[8]); (void)hipEventElapsedTime(&out->ms_dp_main, b->ev[7], b->ev[6]);
          for(int k = 0; k <= DP_LAST_TIER; k++) (void)hipEventElapsedTime(&out->ms_dp_class[k], b->evC[k][0], b->evC[k][1]);
          if(c->jf_grid > 0) (void)hipEventElapsedTime(&out->ms_dp_jump_free, b->evC[0][0], b->evJF);
          if(b->band_used) (void)hipEventElapsedTime(&out->ms_dp_band, b->evBand[0], b->evBand[1]);
          if(b->band2_used) (void)hipEventElapsedTime(&out->ms_dp_band2, b->evBand2[0], b->evBand2[1]);
          if(b->side_used) (void)hipEventElapsedTime(&out->ms_side, b->evSide[1], b->evSide[6]); } }
    { int wc[WC_N]; HIP_TRY(c, hipMemcpyAsync(wc, b->B.work_counter, sizeof(wc), hipMemcpyDeviceToHost, c->active)); HIP_TRY(c, hipStreamSynchronize(c->active)); out->n_chains_retried = 0; for(int k = 1; k <= 6; k++) out->n_chains_retried += wc[12 + 4 * (k - 1)] + wc[14 + 4 * (k - 1)]; out->n_dp_retried_large = wc[28] + wc[30];
      out->n_dp_band = b->band_used ? wc[WC_BAND_CALLS] : 0; out->n_dp_band_failed = b->band_used ? wc[WC_BAND_FAILED] : 0; out->n_dp_jump_free_failed = wc[WC_JF_FAILED];
      out->n_dp_band2 = b->band2_used ? wc[WC_B2_CALLS] : 0; out->n_dp_band2_failed = b->band2_used ? wc[WC_B2_FAILED] : 0;
      out->n_dp_class[0] = wc[8] + wc[9] - out->n_dp_band + out->n_dp_band_failed - out->n_dp_band2 + out->n_dp_band2_failed; out->n_dp_jump_free = c->jf_grid > 0 ? wc[6] : 0; for(int k = 1; k <= 6; k++) out->n_dp_class[k] = wc[12 + 4 * (k - 1)] + wc[14 + 4 * (k - 1)]; }
    if(b->staged & 4) (void)hipEventElapsedTime(&out->ms_pair, b->ev[4], b->ev[5]);
    out->n_chains_extended = (int64_t)cnt[CNT_CHAINS_EXT]; out->n_dp_calls = (int64_t)cnt[CNT_DP_CALLS];
    out->n_dp_iterations = (int64_t)cnt[CNT_DP_ITERS]; out->n_dp_cells = (int64_t)cnt[CNT_DP_CELLS];
    out->n_seed_columns = (int64_t)cnt[CNT_SEED_COLS]; out->n_out_columns = (int64_t)cnt[CNT_OUT_COLS];
    out->n_edges_touched = (int64_t)cnt[CNT_EDGES]; out->n_errors = (int64_t)cnt[CNT_ERRORS]; out->n_dp_shared = (int64_t)cnt[CNT_DP_SHARED];
    return HLALA_OK;
}

}  // extern "C"

// ------------------------------------------------------------------------------------ HLATyper scoring
static int typer_tables(hlala_ctx* c, std::vector<void*>& tmp, TyperTables** out)
{
    TyperTables T;
    double insertionP = c->params.long_read_mode ? 0.075 : 0.001, deletionP = insertionP;          // HLATyper.cpp:935-942
    T.ll_ins_actual = log(insertionP) + log(1.0 / 4.0);                                            // :952-954
    T.ll_deletion = log(deletionP);                                                                // :956
    T.ll_match_mismatch = log(1 - insertionP - deletionP);                                         // :959
    for(int q = 0; q < 256; q++) {
        double pCorrect = host_PhredToPCorrect((unsigned char)(q < 33 ? 33 : q));
        if(pCorrect > 0.999) pCorrect = 0.999;                                                     // veryConservativeReadLikelihoods, :2190-2194
        if(pCorrect == 0) pCorrect = 0.001;                                                        // :2197-2200
        T.ll_match[q] = log(pCorrect);
        double pIncorrect = (1 - pCorrect) * (1.0 / 3.0);
        T.ll_mismatch[q] = log(pIncorrect);
    }
    return dev_upload(c, tmp, &T, 1, out);
}

// The three steps of a locus (hla/HLATyper.cpp:2067-2541) on tables that may stay on the device between them: `keep` (non-null) receives the device tables
// of a step instead of the pool, and a step whose device inputs are given uploads nothing (hlala_type_locus).  Host outputs that are null are not downloaded.
static void hand_over(std::vector<void*>& from, std::vector<void*>* to, void* p)
{
    for(size_t i = 0; i < from.size(); i++) if(from[i] == p) { from.erase(from.begin() + (ptrdiff_t)i); break; }
    to->push_back(p);
}
static int exon_loglik_impl(hlala_ctx* c, const hlala_exon_in* in, double* LL, int32_t* mism, std::vector<void*>* keep, double** dLLout, int** dMout)
{
    if(!c || !in || (!keep && (!LL || !mism))) return HLALA_E_ARG;
    const int C = in->n_clusters, P = in->exon_length, R = in->n_reads;
    if(C < 0 || P < 0 || R < 0) { c->err = "negative sizes"; return HLALA_E_ARG; }
    if(C == 0 || R == 0) return HLALA_OK;
    const size_t npos = (size_t)in->pos_off[R];
    for(size_t i = 0; i < npos; i++) if(in->pos_exon[i] < 0 || in->pos_exon[i] >= P || in->pos_glen[i] < 1) { c->err = "exon position out of range / empty genotype"; return HLALA_E_ARG; }
    std::vector<void*> tmp; int rc = 0;
    auto done = [&](int r) { for(void* p : tmp) pool_release(c, p); return r; };
    TyperTables* dT = nullptr; if((rc = typer_tables(c, tmp, &dT))) return done(rc);
    // cluster sequences transposed to [P][C] on the host side of the upload (one-time per locus)
    std::vector<uint8_t> seqT((size_t)C * P);
    for(int cc = 0; cc < C; cc++) for(int p = 0; p < P; p++) seqT[(size_t)p * C + cc] = in->cluster_seq[(size_t)cc * P + p];
    uint8_t *dSeq, *dG0, *dQ, *dUse; int *dOff, *dExon, *dGlen, *dM; double* dLL;
    if((rc = dev_upload(c, tmp, seqT.data(), seqT.size(), &dSeq))) return done(rc);
    if((rc = dev_upload(c, tmp, in->pos_off, (size_t)R + 1, &dOff))) return done(rc);
    if((rc = dev_upload(c, tmp, in->pos_exon, npos, &dExon))) return done(rc);
    if((rc = dev_upload(c, tmp, in->pos_g0, npos, &dG0))) return done(rc);
    if((rc = dev_upload(c, tmp, in->pos_glen, npos, &dGlen))) return done(rc);
    if((rc = dev_upload(c, tmp, in->pos_qual, npos, &dQ))) return done(rc);
    if((rc = dev_upload(c, tmp, in->pos_use, npos, &dUse))) return done(rc);
    if((rc = dev_alloc(c, tmp, (size_t)C * R, &dLL))) return done(rc);
    if((rc = dev_alloc(c, tmp, (size_t)C * R, &dM))) return done(rc);
    hipLaunchKernelGGL(k_exon_loglik, dim3((C + 255) / 256, R), dim3(256), 0, c->active, dT, C, P, R, dSeq, dOff, dExon, dG0, dGlen, dQ, dUse, dLL, dM);
    if((rc = check_launch(c, "k_exon_loglik"))) return done(rc);
    if((LL && hipMemcpyAsync(LL, dLL, (size_t)C * R * 8, hipMemcpyDeviceToHost, c->active) != hipSuccess) || (mism && hipMemcpyAsync(mism, dM, (size_t)C * R * 4, hipMemcpyDeviceToHost, c->active) != hipSuccess) ||
       hipStreamSynchronize(c->active) != hipSuccess) { c->err = "hlala_exon_loglik: download failed"; return done(HLALA_E_DEVICE); }       // (synchronised either way: the host arrays uploaded above are the caller's)
    if(keep) { hand_over(tmp, keep, dLL); hand_over(tmp, keep, dM); *dLLout = dLL; *dMout = dM; }
    return done(HLALA_OK);
}
extern "C" int hlala_exon_loglik(hlala_ctx* c, const hlala_exon_in* in, double* LL, int32_t* mism)
{
    DEV_GUARD(c);
    ReaderScope rscope_(c, nullptr);
    if(!c || !in || !LL || !mism) return HLALA_E_ARG;
    return exon_loglik_impl(c, in, LL, mism, nullptr, nullptr, nullptr);
}

static int pair_loglik_impl(hlala_ctx* c, const double* LL, const int32_t* mism, const double* dLLin, const int* dMin, int32_t C, int32_t R, double* pairLL, double* misAvg, double* misMin,
                           std::vector<void*>* keep, double** dPout, double** dAout, double** dMnout)
{
    if(!c || (!dLLin && (!LL || !mism)) || !pairLL || !misAvg || !misMin || C < 0 || R < 0) return HLALA_E_ARG;
    if(C == 0) return HLALA_OK;
    std::vector<void*> tmp; int rc = 0;
    auto done = [&](int r) { for(void* p : tmp) pool_release(c, p); return r; };
    const size_t npairs = (size_t)C * (C + 1) / 2, nCR = (size_t)C * R;
    double *dLL, *dLLT, *dP, *dA, *dMn; int *dM, *dMT;
    if(dLLin) { dLL = const_cast<double*>(dLLin); dM = const_cast<int*>(dMin); }
    else {
        if((rc = dev_upload(c, tmp, LL, nCR, &dLL))) return done(rc);
        if((rc = dev_upload(c, tmp, mism, nCR, &dM))) return done(rc);
    }
    if((rc = dev_alloc(c, tmp, nCR, &dLLT))) return done(rc);
    if((rc = dev_alloc(c, tmp, nCR, &dMT))) return done(rc);
    if((rc = dev_alloc(c, tmp, npairs, &dP))) return done(rc);
    if((rc = dev_alloc(c, tmp, npairs, &dA))) return done(rc);
    if((rc = dev_alloc(c, tmp, npairs, &dMn))) return done(rc);
    if(R > 0) {
        dim3 tb(32, 32), tg((R + 31) / 32, (C + 31) / 32);
        hipLaunchKernelGGL(k_transpose<double>, tg, tb, 0, c->active, C, R, dLL, dLLT);
        hipLaunchKernelGGL(k_transpose<int>, tg, tb, 0, c->active, C, R, dM, dMT);
    }
    hipLaunchKernelGGL(k_pair_loglik, dim3((C + 255) / 256, (C + PAIRLL_ROWS - 1) / PAIRLL_ROWS), dim3(256), 0, c->active, C, R, dLL, dLLT, dM, dMT, dP, dA, dMn);
    if((rc = check_launch(c, "k_pair_loglik"))) return done(rc);
    if(hipMemcpyAsync(pairLL, dP, npairs * 8, hipMemcpyDeviceToHost, c->active) != hipSuccess || hipMemcpyAsync(misAvg, dA, npairs * 8, hipMemcpyDeviceToHost, c->active) != hipSuccess ||
       hipMemcpyAsync(misMin, dMn, npairs * 8, hipMemcpyDeviceToHost, c->active) != hipSuccess || (!keep && hipStreamSynchronize(c->active) != hipSuccess)) { c->err = "hlala_pair_loglik: download failed"; return done(HLALA_E_DEVICE); }
    if(keep) { hand_over(tmp, keep, dP); hand_over(tmp, keep, dA); hand_over(tmp, keep, dMn); *dPout = dP; *dAout = dA; *dMnout = dMn;
               hand_over(tmp, keep, dLLT); hand_over(tmp, keep, dMT); }      // (not synchronised: the caller's next step runs behind this one on the same stream; its scratch stays allocated until then)
    return done(HLALA_OK);
}
extern "C" int hlala_pair_loglik(hlala_ctx* c, const double* LL, const int32_t* mism, int32_t C, int32_t R, double* pairLL, double* misAvg, double* misMin)
{
    DEV_GUARD(c);
    ReaderScope rscope_(c, nullptr);
    if(!c || !LL || !mism) return HLALA_E_ARG;
    return pair_loglik_impl(c, LL, mism, nullptr, nullptr, C, R, pairLL, misAvg, misMin, nullptr, nullptr, nullptr, nullptr);
}

// ---- known-answer kernels
namespace hlala {
__global__ void k_kat_phred(const DevTables* T, int n, const double* p, uint8_t* out)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if(i < n) out[i] = phred_from_pcorrect(*T, p[i]);
}
__global__ void k_kat_rand(int n, u32* seeds, int* vals)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if(i < n) { unsigned int s = seeds[i]; vals[i] = glibc_rand_r(&s); seeds[i] = s; }
}
__global__ void k_kat_exp(int n, const double* x, double* y)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if(i < n) y[i] = exp_cr_nonpos(x[i]);
}
}  // namespace hlala

// HLALA_DEBUG=1: the debug buffer of the context (8192 ints in device memory, copied out here); clear != 0 zeroes it afterwards
extern "C" int hlala_debug_buffer(hlala_ctx* c, int* out8192, int clear)
{
    if(!c || !c->dbg_host) return HLALA_E_STATE;
    DEV_GUARD(c);
    HIP_TRY(c, hipStreamSynchronize(c->active));
    if(out8192) HIP_TRY(c, hipMemcpy(out8192, c->dbg_host, 8192 * sizeof(int), hipMemcpyDeviceToHost));
    if(clear) HIP_TRY(c, hipMemset(c->dbg_host, 0, 8192 * sizeof(int)));
    return HLALA_OK;
}

extern "C" int hlala_debug_counters(hlala_ctx* c, hlala_batch* b, unsigned long long* out32)
{
    DEV_GUARD(c);
    ReaderScope rscope_(c, b); if(rscope_.rc) return rscope_.rc;
    if(!c || !b || !out32) return HLALA_E_ARG;
    if(!b->outputs_ready) { memset(out32, 0, 32 * sizeof(u64)); return HLALA_OK; }      // only uploaded so far: the counters do not exist yet
    HIP_TRY(c, hipStreamSynchronize(c->active));
    HIP_TRY(c, hipMemcpyAsync(out32, b->B.counters, 32 * sizeof(u64), hipMemcpyDeviceToHost, c->active)); HIP_TRY(c, hipStreamSynchronize(c->active));
    return HLALA_OK;
}

// device memory of a batch and of its context (diagnostics; bench.py reports them): out[0] = bytes of the batch's blocks (inputs + outputs, pool size classes included),
// out[1] = chains that hold column rows (batch.h: chain_row; n_chains when every chain does), out[2] = bytes parked in the context's pool, out[3] = bytes of every block the
// context has handed out or parked (graph, tables, slabs, batches)
extern "C" int hlala_debug_memory(hlala_ctx* c, hlala_batch* b, unsigned long long* out4)
{
    if(!c || !out4) return HLALA_E_ARG;
    out4[0] = out4[1] = 0;
    if(b) {
        for(void* p : b->allocs) { auto it = c->block_bytes.find(p); if(it != c->block_bytes.end()) out4[0] += it->second; }
        out4[1] = (unsigned long long)(b->prepared ? b->B.n_rows : b->B.n_chains);
    }
    out4[2] = c->pool_bytes; out4[3] = 0;
    for(auto& kv : c->block_bytes) out4[3] += kv.second;
    return HLALA_OK;
}

// the batch's work counters (diagnostics: item counts of the lists, fail-over reasons of the band kernel -- batch.h)
extern "C" int hlala_debug_work_counters(hlala_ctx* c, hlala_batch* b, int* out, int capacity)
{
    static_assert(WC_N == HLALA_DEBUG_WC_N && WC_BAND_FETCH == HLALA_DEBUG_WC_BAND_FETCH && WC_BAND_WHY == HLALA_DEBUG_WC_BAND_WHY && WC_BAND_TIED == HLALA_DEBUG_WC_BAND_TIED, "include/hlala_gpu.h: debug section");
    DEV_GUARD(c);
    ReaderScope rscope_(c, b); if(rscope_.rc) return rscope_.rc;
    if(!c || !b || !out || capacity < 0) return HLALA_E_ARG;
    const int n = capacity < WC_N ? capacity : WC_N;          // (a caller built against an older header gets the counters it has room for)
    if(!b->outputs_ready) { memset(out, 0, (size_t)n * sizeof(int)); return HLALA_OK; }
    HIP_TRY(c, hipStreamSynchronize(c->active));
    HIP_TRY(c, hipMemcpyAsync(out, b->B.work_counter, (size_t)n * sizeof(int), hipMemcpyDeviceToHost, c->active)); HIP_TRY(c, hipStreamSynchronize(c->active));
    return HLALA_OK;
}

// diagnostics (tools/tier_predict.py): the DP items of the batch's last extension stage ([2 * n_chains] records of 8 ints: item, read offset, read length, start offset,
// start level, start node, class, linear run; item < 0: no call) and its retry lists ([16 * n_chains] slots of dp_items; counts in the work counters)
extern "C" int hlala_debug_dp_items(hlala_ctx* c, hlala_batch* b, int* items_out, long long items_capacity_bytes, int* retry_out, long long retry_capacity_bytes)
{
    DEV_GUARD(c);
    ReaderScope rscope_(c, b); if(rscope_.rc) return rscope_.rc;
    if(!c || !b) return HLALA_E_ARG;
    if(!b->outputs_ready || !(b->staged & 2)) { c->err = "hlala_debug_dp_items before the extension stage"; return HLALA_E_STATE; }
    const size_t nc = (size_t)b->B.n_chains;
    if((items_out && items_capacity_bytes < (long long)(2 * nc * 32)) || (retry_out && retry_capacity_bytes < (long long)(16 * nc * sizeof(int)))) {
        c->err = "hlala_debug_dp_items: buffer too small (items 64 bytes per chain, retry lists 64 bytes per chain)"; return HLALA_E_ARG; }
    if(items_out) HIP_TRY(c, hipMemcpyAsync(items_out, b->B.dp_items, 2 * nc * 32, hipMemcpyDeviceToHost, c->active));
    if(retry_out) HIP_TRY(c, hipMemcpyAsync(retry_out, b->B.retry_list, 16 * nc * sizeof(int), hipMemcpyDeviceToHost, c->active));
    HIP_TRY(c, hipStreamSynchronize(c->active));
    return HLALA_OK;
}

extern "C" int hlala_kat_phred(hlala_ctx* c, int n, const double* p_correct, uint8_t* phred_out, const uint8_t* phred_in, double* p_out)
{
    DEV_GUARD(c);
    if(!c || n < 0) return HLALA_E_ARG;
    if(p_correct && phred_out && n) {
        double* dp = nullptr; uint8_t* dq = nullptr; std::vector<void*> tmp;
        int rc = dev_upload(c, tmp, p_correct, (size_t)n, &dp); if(rc) return rc;
        rc = dev_alloc(c, tmp, (size_t)n, &dq); if(rc) return rc;
        hipLaunchKernelGGL(k_kat_phred, dim3((n + 255) / 256), dim3(256), 0, c->active, c->dT, n, dp, dq);
        HIP_TRY(c, hipMemcpyAsync(phred_out, dq, (size_t)n, hipMemcpyDeviceToHost, c->active));
        HIP_TRY(c, hipStreamSynchronize(c->active));
        for(void* p : tmp) pool_release(c, p);
    }
    if(phred_in && p_out) {
        // PhredToPCorrect feeds the host-built likelihood tables: report the table entries' pre-image
        for(int i = 0; i < n; i++) p_out[i] = host_PhredToPCorrect(phred_in[i]);
    }
    return HLALA_OK;
}

static int call_locus_impl(hlala_ctx* c, int32_t C, const double* pairLL, const double* misAvg, const double* misMin, const double* dLLin, const double* dMAin, const double* dMMin,
                          int32_t* order, double* p_normalized, double* cluster_marginal, hlala_call_out* out)
{
    if(!c || C < 1 || (!dLLin && (!pairLL || !misAvg || !misMin)) || !out) return HLALA_E_ARG;
    if(C > 46000) { c->err = "hlala_call_locus: more than 46000 clusters (pair index exceeds 31 bits)"; return HLALA_E_CAPACITY; }
    const long long nP = (long long)C * (C + 1) / 2, n2 = 2 * nP;
    std::vector<void*> tmp;
    auto done = [&](int r_) { for(void* p : tmp) pool_release(c, p); return r_; };       // scratch is parked for the next locus
    int rc = 0;
    double *dLL = nullptr, *dMA = nullptr, *dMM = nullptr, *dP = nullptr, *dPart = nullptr, *dScal = nullptr, *dVal = nullptr, *dVal2 = nullptr, *dMarg = nullptr;
    u64 *dK1 = nullptr, *dK2 = nullptr, *dCK = nullptr, *dCK2 = nullptr; int *dI1 = nullptr, *dI2 = nullptr, *dC1 = nullptr, *dC2 = nullptr, *dTies = nullptr;
    long long* dPidx = nullptr; hlala_call_out* dOut = nullptr;
    const int NB = 1024;
    if(dLLin) { dLL = const_cast<double*>(dLLin); dMA = const_cast<double*>(dMAin); dMM = const_cast<double*>(dMMin); }
    else if((rc = dev_upload(c, tmp, pairLL, (size_t)nP, &dLL)) || (rc = dev_upload(c, tmp, misAvg, (size_t)nP, &dMA)) || (rc = dev_upload(c, tmp, misMin, (size_t)nP, &dMM))) return done(rc);
    if((rc = dev_alloc(c, tmp, (size_t)nP, &dP)) || (rc = dev_alloc(c, tmp, (size_t)NB, &dPart)) || (rc = dev_alloc(c, tmp, (size_t)NB, &dPidx)) || (rc = dev_alloc(c, tmp, 4, &dScal)) ||
       (rc = dev_alloc(c, tmp, (size_t)nP, &dK1)) || (rc = dev_alloc(c, tmp, (size_t)nP, &dK2)) || (rc = dev_alloc(c, tmp, (size_t)nP, &dI1)) || (rc = dev_alloc(c, tmp, (size_t)nP, &dI2)) ||
       (rc = dev_alloc(c, tmp, (size_t)nP, &dC1)) || (rc = dev_alloc(c, tmp, (size_t)nP, &dC2)) || (rc = dev_alloc(c, tmp, (size_t)n2, &dCK)) || (rc = dev_alloc(c, tmp, (size_t)n2, &dCK2)) ||
       (rc = dev_alloc(c, tmp, (size_t)n2, &dVal)) || (rc = dev_alloc(c, tmp, (size_t)n2, &dVal2)) || (rc = dev_alloc(c, tmp, (size_t)C, &dMarg)) || (rc = dev_alloc(c, tmp, 1, &dTies, true)) ||
       (rc = dev_alloc(c, tmp, 1, &dOut))) return done(rc);
    long long* dMaxIdx = (long long*)(dScal + 2);      // dScal[0] = LL max, dScal[1] = P sum, dScal[2..3] as one long long = index of the maximum
    hipStream_t st = c->active;
    const int T = 256; const unsigned gP = (unsigned)((nP + T - 1) / T);
    hipLaunchKernelGGL(k_call_max, dim3(NB), dim3(T), 0, st, dLL, nP, dPart, dPidx);
    hipLaunchKernelGGL(k_call_max_final, dim3(1), dim3(1), 0, st, dPart, dPidx, NB, dScal, dMaxIdx);
    hipLaunchKernelGGL(k_call_p, dim3(NB), dim3(T), 0, st, dLL, nP, dScal, dP, dPart);
    hipLaunchKernelGGL(k_call_psum_final, dim3(1), dim3(1), 0, st, dPart, NB, dScal + 1);
    hipLaunchKernelGGL(k_call_normalize, dim3(gP), dim3(T), 0, st, dP, nP, dScal + 1);
    // order: stable sort by Mism_avg ascending, then stable sort by LL descending
    hipLaunchKernelGGL(k_call_keys_mism, dim3(gP), dim3(T), 0, st, dMA, nP, dK1, dI1);
    size_t cubBytes = 0, cubBytes2 = 0;
    HIP_TRY_F(c, hipcub::DeviceRadixSort::SortPairs(nullptr, cubBytes, dK1, dK2, dI1, dI2, (int)nP, 0, 64, st), done);
    HIP_TRY_F(c, hipcub::DeviceRadixSort::SortPairs(nullptr, cubBytes2, dCK, dCK2, dVal, dVal2, (int)n2, 0, 64, st), done);
    if(cubBytes2 > cubBytes) cubBytes = cubBytes2;
    char* dCub = nullptr; if((rc = dev_alloc(c, tmp, cubBytes ? cubBytes : 1, &dCub))) return done(rc);
    HIP_TRY_F(c, hipcub::DeviceRadixSort::SortPairs(dCub, cubBytes, dK1, dK2, dI1, dI2, (int)nP, 0, 64, st), done);
    hipLaunchKernelGGL(k_call_keys_ll, dim3(gP), dim3(T), 0, st, dLL, dI2, nP, dK1);
    HIP_TRY_F(c, hipcub::DeviceRadixSort::SortPairs(dCub, cubBytes, dK1, dK2, dI2, dI1, (int)nP, 0, 64, st), done);       // dI1 = order
    // marginals in the reference's accumulation order
    hipLaunchKernelGGL(k_call_clusters, dim3((unsigned)C), dim3(128), 0, st, (int)C, dC1, dC2);
    hipLaunchKernelGGL(k_call_contrib, dim3(gP), dim3(T), 0, st, dI1, dC1, dC2, dP, nP, dCK, dVal, dTies, dLL, dMA);
    HIP_TRY_F(c, hipcub::DeviceRadixSort::SortPairs(dCub, cubBytes, dCK, dCK2, dVal, dVal2, (int)n2, 0, 64, st), done);
    hipLaunchKernelGGL(k_call_marginals, dim3((unsigned)((C + 63) / 64)), dim3(64), 0, st, dCK2, dVal2, n2, (int)C, dMarg);
    hipLaunchKernelGGL(k_call_decide, dim3(1), dim3(64), 0, st, (int)C, dMarg, dP, dMM, dScal, dMaxIdx, dTies, dOut);
    rc = check_launch(c, "hlala_call_locus kernels"); if(rc) return done(rc);
    if((rc = dl(c, order, dI1, (size_t)nP)) || (rc = dl(c, p_normalized, dP, (size_t)nP)) || (rc = dl(c, cluster_marginal, dMarg, (size_t)C)) || (rc = dl(c, out, dOut, 1))) return done(rc);
    HIP_TRY_F(c, hipStreamSynchronize(st), done);
    return done(HLALA_OK);
}
extern "C" int hlala_call_locus(hlala_ctx* c, int32_t C, const double* pairLL, const double* misAvg, const double* misMin,
                                int32_t* order, double* p_normalized, double* cluster_marginal, hlala_call_out* out)
{
    DEV_GUARD(c);
    ReaderScope rscope_(c, nullptr);
    if(!c || !pairLL || !misAvg || !misMin) return HLALA_E_ARG;
    return call_locus_impl(c, C, pairLL, misAvg, misMin, nullptr, nullptr, nullptr, order, p_normalized, cluster_marginal, out);
}

// hlala_exon_loglik -> hlala_pair_loglik -> hlala_call_locus with the tables left on the device in between: the per-read table (clusters x reads) is neither
// downloaded nor uploaded again, the all-pairs tables go down once (the caller's files need them) and are not uploaded for the call.
extern "C" int hlala_type_locus(hlala_ctx* c, const hlala_exon_in* in, double* LL, int32_t* mism, double* pairLL, double* misAvg, double* misMin,
                                int32_t* order, double* p_normalized, double* cluster_marginal, hlala_call_out* out)
{
    DEV_GUARD(c);
    ReaderScope rscope_(c, nullptr);
    if(!c || !in || !pairLL || !misAvg || !misMin || !order || !p_normalized || !cluster_marginal || !out) return HLALA_E_ARG;
    if(in->n_clusters < 1) { c->err = "hlala_type_locus: no clusters"; return HLALA_E_ARG; }
    std::vector<void*> keep;
    // (the steps run unsynchronised in keep mode: on ANY failure the stream is drained before the kept device blocks go back to the pool and before the caller, who
    //  sees the error, may free the host buffers the queued copies write to)
    auto done = [&](int r_) { if(r_) (void)hipStreamSynchronize(c->active); for(void* p : keep) pool_release(c, p); return r_; };
    const int C = in->n_clusters, R = in->n_reads < 0 ? 0 : in->n_reads;
    double *dLL = nullptr, *dP = nullptr, *dA = nullptr, *dMn = nullptr; int* dM = nullptr;
    int rc = exon_loglik_impl(c, in, LL, mism, &keep, &dLL, &dM); if(rc) return done(rc);
    if(!dLL) {                                                     // no reads at the locus: empty per-read tables (hlala_exon_loglik writes nothing either)
        if((rc = dev_alloc(c, keep, 1, &dLL)) || (rc = dev_alloc(c, keep, 1, &dM))) return done(rc);
    }
    if((rc = pair_loglik_impl(c, nullptr, nullptr, dLL, dM, C, R, pairLL, misAvg, misMin, &keep, &dP, &dA, &dMn))) return done(rc);
    rc = call_locus_impl(c, C, nullptr, nullptr, nullptr, dP, dA, dMn, order, p_normalized, cluster_marginal, out);       // (ends synchronised)
    return done(rc);
}

extern "C" int hlala_exon_positions(hlala_ctx* c, hlala_batch* b, const hlala_locus_desc* L, hlala_exon_positions_out* o)
{
    DEV_GUARD(c);
    ReaderScope rscope_(c, b); if(rscope_.rc) return rscope_.rc;
    if(!c || !b || !L || !o) return HLALA_E_ARG;
    if(!(b->staged & 4)) { c->err = "hlala_exon_positions before hlala_pair_chains"; return HLALA_E_STATE; }
    if(L->level_max < L->level_min || !L->level_to_exon) { c->err = "locus: empty level range or no level_to_exon table"; return HLALA_E_ARG; }
    DevBatch& B = b->B;
    const int np = B.n_pairs;
    o->n_reads = o->n_pos = o->n_chars = 0; o->n_pairs_ok = o->n_pairs_broken = 0;
    if(np <= 0) { if(o->pos_off && o->cap_reads >= 0) o->pos_off[0] = 0; if(o->geno_off && o->cap_pos >= 0) o->geno_off[0] = 0; return HLALA_OK; }
    std::vector<void*> tmp;
    auto done = [&](int r_) { for(void* p : tmp) pool_release(c, p); return r_; };
    int rc = 0;
    ExonLocus EL; EL.level_min = L->level_min; EL.level_max = L->level_max; EL.insert_mean = L->insert_mean; EL.insert_sd = L->insert_sd;
    EL.min_mapq = L->min_mapq; EL.min_weighted_ok = L->min_weighted_ok; EL.level_to_exon = nullptr; EL.pair_mask = nullptr; EL.min_alignment_columns = L->min_alignment_columns;
    int* dL2E = nullptr; uint8_t* dMask = nullptr; int *dCnt = nullptr, *dOff = nullptr, *dOB = nullptr; char* dCub = nullptr;
    if((rc = dev_upload(c, tmp, L->level_to_exon, (size_t)(L->level_max - L->level_min + 1), &dL2E))) return done(rc);
    EL.level_to_exon = dL2E;
    if(L->pair_mask) { if((rc = dev_upload(c, tmp, L->pair_mask, (size_t)np, &dMask))) return done(rc); EL.pair_mask = dMask; }
    if((rc = dev_alloc(c, tmp, (size_t)3 * np + 3, &dCnt)) || (rc = dev_alloc(c, tmp, (size_t)3 * np + 3, &dOff)) || (rc = dev_alloc(c, tmp, 2, &dOB, true))) return done(rc);
    hipStream_t st = c->active;
    hlala_exon_positions_out dO; memset(&dO, 0, sizeof(dO));
    const unsigned grid = (unsigned)((np + 127) / 128);
    hipLaunchKernelGGL((k_exon_positions<0>), dim3(grid), dim3(128), 0, st, b->dB, c->dT, EL, dCnt, (const int*)nullptr, dOB, dO);
    if((rc = check_launch(c, "k_exon_positions<0>"))) return done(rc);
    // interleaved (read, positions, characters) counts -> three exclusive sums in one scan over 3-vectors: scan each column separately
    // with a strided view is not what the library offers, so the counts are de-interleaved by scanning the flat array three times with
    // a transform; simpler and still tiny: one scan over the flat array of a struct of three ints
    struct I3 { int a, b, c; };
    struct I3Add { __host__ __device__ I3 operator()(const I3& x, const I3& y) const { I3 r; r.a = x.a + y.a; r.b = x.b + y.b; r.c = x.c + y.c; return r; } };
    size_t cubBytes = 0; I3 zero; zero.a = zero.b = zero.c = 0;
    HIP_TRY_F(c, hipcub::DeviceScan::ExclusiveScan(nullptr, cubBytes, (I3*)dCnt, (I3*)dOff, I3Add(), zero, np + 1, st), done);
    if((rc = dev_alloc(c, tmp, cubBytes ? cubBytes : 1, &dCub))) return done(rc);
    HIP_TRY_F(c, hipMemsetAsync(dCnt + 3 * (size_t)np, 0, 3 * sizeof(int), st), done);
    HIP_TRY_F(c, hipcub::DeviceScan::ExclusiveScan(dCub, cubBytes, (I3*)dCnt, (I3*)dOff, I3Add(), zero, np + 1, st), done);
    int totals[3] = {0, 0, 0}, ob[2] = {0, 0};
    HIP_TRY_F(c, hipMemcpyAsync(totals, dOff + 3 * (size_t)np, sizeof(totals), hipMemcpyDeviceToHost, st), done);
    HIP_TRY_F(c, hipMemcpyAsync(ob, dOB, sizeof(ob), hipMemcpyDeviceToHost, st), done);
    HIP_TRY_F(c, hipStreamSynchronize(st), done);
    o->n_reads = totals[0]; o->n_pos = totals[1]; o->n_chars = totals[2]; o->n_pairs_ok = ob[0]; o->n_pairs_broken = ob[1];
    if(totals[0] > o->cap_reads || totals[1] > o->cap_pos || totals[2] > o->cap_chars) { c->err = "hlala_exon_positions: output capacity too small (needed sizes are in n_reads / n_pos / n_chars)"; return done(HLALA_E_CAPACITY); }
    const size_t nR = (size_t)totals[0], nPz = (size_t)totals[1], nC = (size_t)totals[2];
    if((rc = dev_alloc(c, tmp, nR, &dO.read_pair)) || (rc = dev_alloc(c, tmp, 2 * nR, &dO.read_weighted_ok)) || (rc = dev_alloc(c, tmp, 2 * nR, &dO.read_fraction_ok)) ||
       (rc = dev_alloc(c, tmp, nR, &dO.read_distance)) || (rc = dev_alloc(c, tmp, 2 * nR, &dO.read_cols_nongap)) || (rc = dev_alloc(c, tmp, nR + 1, &dO.pos_off)) ||
       (rc = dev_alloc(c, tmp, nPz, &dO.pos_exon)) || (rc = dev_alloc(c, tmp, nPz, &dO.pos_level)) || (rc = dev_alloc(c, tmp, nPz, &dO.pos_mate)) || (rc = dev_alloc(c, tmp, nPz, &dO.pos_mapq)) ||
       (rc = dev_alloc(c, tmp, nPz, &dO.pos_novel_gap)) || (rc = dev_alloc(c, tmp, nPz + 1, &dO.geno_off)) || (rc = dev_alloc(c, tmp, nC, &dO.geno_chars)) || (rc = dev_alloc(c, tmp, nC, &dO.qual_chars)) ||
       (rc = dev_alloc(c, tmp, 2 * nR, &dO.read_reverse)) || (rc = dev_alloc(c, tmp, 2 * nR, &dO.read_mapq))) return done(rc);
    hipLaunchKernelGGL((k_exon_positions<1>), dim3(grid), dim3(128), 0, st, b->dB, c->dT, EL, dCnt, (const int*)dOff, dOB, dO);
    if((rc = check_launch(c, "k_exon_positions<1>"))) return done(rc);
    if((rc = dl(c, o->read_pair, dO.read_pair, nR)) || (rc = dl(c, o->read_weighted_ok, dO.read_weighted_ok, 2 * nR)) || (rc = dl(c, o->read_fraction_ok, dO.read_fraction_ok, 2 * nR)) ||
       (rc = dl(c, o->read_distance, dO.read_distance, nR)) || (rc = dl(c, o->read_cols_nongap, dO.read_cols_nongap, 2 * nR)) || (rc = dl(c, o->pos_off, dO.pos_off, nR)) ||
       (rc = dl(c, o->pos_exon, dO.pos_exon, nPz)) || (rc = dl(c, o->pos_level, dO.pos_level, nPz)) || (rc = dl(c, o->pos_mate, dO.pos_mate, nPz)) || (rc = dl(c, o->pos_mapq, dO.pos_mapq, nPz)) ||
       (rc = dl(c, o->pos_novel_gap, dO.pos_novel_gap, nPz)) || (rc = dl(c, o->geno_off, dO.geno_off, nPz)) || (rc = dl(c, o->geno_chars, dO.geno_chars, nC)) || (rc = dl(c, o->qual_chars, dO.qual_chars, nC)) ||
       (rc = dl(c, o->read_reverse, dO.read_reverse, 2 * nR)) || (rc = dl(c, o->read_mapq, dO.read_mapq, 2 * nR))) return done(rc);
    HIP_TRY_F(c, hipStreamSynchronize(st), done);
    if(o->pos_off) o->pos_off[nR] = (int32_t)nPz;
    if(o->geno_off) o->geno_off[nPz] = (int32_t)nC;
    return done(HLALA_OK);
}

extern "C" int hlala_unit_alignment_stats(hlala_ctx* c, hlala_batch* b, hlala_unit_stats_out* o)
{
    DEV_GUARD(c);
    ReaderScope rscope_(c, b); if(rscope_.rc) return rscope_.rc;
    if(!c || !b || !o || !o->valid || !o->strands_valid || !o->distance || !o->fraction_ok || !o->weighted_ok || !o->n_columns || !o->mate_mapq) return HLALA_E_ARG;
    if(!(b->staged & 4)) { c->err = "hlala_unit_alignment_stats before hlala_pair_chains"; return HLALA_E_STATE; }
    const size_t n = (size_t)b->B.n_pairs;
    if(n == 0) return HLALA_OK;
    std::vector<void*> tmp;
    auto done = [&](int r_) { for(void* p : tmp) pool_release(c, p); return r_; };
    int rc = 0; hlala_unit_stats_out d; memset(&d, 0, sizeof(d));
    if((rc = dev_alloc(c, tmp, n, &d.valid)) || (rc = dev_alloc(c, tmp, n, &d.strands_valid)) || (rc = dev_alloc(c, tmp, n, &d.distance)) || (rc = dev_alloc(c, tmp, 2 * n, &d.fraction_ok)) ||
       (rc = dev_alloc(c, tmp, 2 * n, &d.weighted_ok)) || (rc = dev_alloc(c, tmp, 2 * n, &d.n_columns)) || (rc = dev_alloc(c, tmp, 2 * n, &d.mate_mapq))) return done(rc);
    hipLaunchKernelGGL(k_unit_stats, dim3((unsigned)((n + 127) / 128)), dim3(128), 0, c->active, b->dB, c->dT, d);
    if((rc = check_launch(c, "k_unit_stats"))) return done(rc);
    if((rc = dl(c, o->valid, d.valid, n)) || (rc = dl(c, o->strands_valid, d.strands_valid, n)) || (rc = dl(c, o->distance, d.distance, n)) || (rc = dl(c, o->fraction_ok, d.fraction_ok, 2 * n)) ||
       (rc = dl(c, o->weighted_ok, d.weighted_ok, 2 * n)) || (rc = dl(c, o->n_columns, d.n_columns, 2 * n)) || (rc = dl(c, o->mate_mapq, d.mate_mapq, 2 * n))) return done(rc);
    HIP_TRY_F(c, hipStreamSynchronize(c->active), done);
    return done(HLALA_OK);
}

// canonical codes of the query k-mers and the sorted set of the distinct ones; a query with a character outside ACGT cannot occur in a read k-mer over ACGT
static int kmer_queries(hlala_ctx* c, const char* who, int32_t k, int32_t n_queries, const char* queries, std::vector<u64>& canon, std::vector<u64>& uniq)
{
    if(k < 1 || k > 31) { c->err = std::string(who) + ": k must be in 1..31 (2-bit codes in one 64-bit word)"; return HLALA_E_ARG; }
    canon.assign((size_t)n_queries, ~0ull); uniq.clear();
    for(int i = 0; i < n_queries; i++) {
        u64 f = 0, rc = 0; bool ok = true;
        for(int j = 0; j < k; j++) {
            const char ch = queries[(size_t)i * k + j];
            const int cj = ch == 'A' ? 0 : ch == 'C' ? 1 : ch == 'G' ? 2 : ch == 'T' ? 3 : 4;
            if(cj > 3) { ok = false; break; }
            f = (f << 2) | (u64)cj; rc |= (u64)(3 - cj) << (2 * j);
        }
        if(ok) { canon[i] = rc < f ? rc : f; uniq.push_back(canon[i]); }
    }
    std::sort(uniq.begin(), uniq.end()); uniq.erase(std::unique(uniq.begin(), uniq.end()), uniq.end());
    if(uniq.size() > (size_t)hlala::KMER_QCAP) { c->err = std::string(who) + ": more than 4096 distinct query k-mers in one call"; return HLALA_E_CAPACITY; }
    return HLALA_OK;
}

extern "C" int hlala_kmer_presence(hlala_ctx* c, hlala_batch* b, const uint8_t* pair_mask, int32_t k, int32_t n_queries, const char* queries, uint8_t* present)
{
    DEV_GUARD(c);
    ReaderScope rscope_(c, b); if(rscope_.rc) return rscope_.rc;
    if(!c || !b || n_queries < 0 || (n_queries > 0 && (!queries || !present))) return HLALA_E_ARG;
    std::vector<u64> canon, uniq;
    int rc = kmer_queries(c, "hlala_kmer_presence", k, n_queries, queries, canon, uniq); if(rc) return rc;
    if(n_queries == 0) return HLALA_OK;
    memset(present, 0, (size_t)n_queries);
    if(uniq.empty() || b->B.n_pairs <= 0) return HLALA_OK;
    std::vector<void*> tmp;
    auto done = [&](int r_) { for(void* p : tmp) pool_release(c, p); return r_; };
    u64* dQ = nullptr; uint8_t *dP = nullptr, *dMask = nullptr;
    if((rc = dev_upload(c, tmp, uniq.data(), uniq.size(), &dQ)) || (rc = dev_alloc(c, tmp, uniq.size(), &dP, true))) return done(rc);
    if(pair_mask && (rc = dev_upload(c, tmp, pair_mask, (size_t)b->B.n_pairs, &dMask))) return done(rc);
    const int nReads = b->B.unpaired ? b->B.n_pairs : 2 * b->B.n_pairs;
    const unsigned grid = (unsigned)std::min<long long>((long long)nReads, (long long)c->stitch_grid);
    hipLaunchKernelGGL(k_kmer_presence, dim3(grid), dim3(64), 0, c->active, b->dB, (const uint8_t*)dMask, (int)k, (int)uniq.size(), (const u64*)dQ, dP);
    if((rc = check_launch(c, "k_kmer_presence"))) return done(rc);
    std::vector<uint8_t> hp(uniq.size());
    if((rc = dl(c, hp.data(), dP, uniq.size()))) return done(rc);
    HIP_TRY_F(c, hipStreamSynchronize(c->active), done);
    for(int i = 0; i < n_queries; i++) if(canon[i] != ~0ull) present[i] = hp[(size_t)(std::lower_bound(uniq.begin(), uniq.end(), canon[i]) - uniq.begin())];
    return done(HLALA_OK);
}

extern "C" void hlala_kmer_forget_reads(hlala_ctx* c)
{
    if(!c) return;
    DEV_GUARD(c);
    if(!c->kept.empty() && c->rs) (void)hipStreamSynchronize(c->rs);
    for(hlala_ctx::KeptReads& kr : c->kept) { pool_release(c, kr.store); pool_release(c, kr.start); pool_release(c, kr.length); }
    c->kept.clear();
}

extern "C" int hlala_kmer_keep_reads(hlala_ctx* c, hlala_batch* b, const uint8_t* pair_mask, int64_t* n_reads_kept)
{
    DEV_GUARD(c);
    ReaderScope rscope_(c, b); if(rscope_.rc) return rscope_.rc;
    if(!c || !b) return HLALA_E_ARG;
    if(n_reads_kept) *n_reads_kept = 0;
    if(b->B.n_pairs <= 0) return HLALA_OK;
    std::vector<void*> tmp;
    auto done = [&](int r_) { for(void* p : tmp) pool_release(c, p); return r_; };
    int rc = 0; uint8_t* dMask = nullptr; unsigned long long* dTot = nullptr;
    if((rc = dev_alloc(c, tmp, 4, &dTot, true))) return done(rc);
    if(pair_mask && (rc = dev_upload(c, tmp, pair_mask, (size_t)b->B.n_pairs, &dMask))) return done(rc);
    const int nReads = b->B.unpaired ? b->B.n_pairs : 2 * b->B.n_pairs;
    hipLaunchKernelGGL(k_kmer_count_kept, dim3((unsigned)std::min<long long>(((long long)nReads + 255) / 256, 4096)), dim3(256), 0, c->active, b->dB, (const uint8_t*)dMask, dTot);
    if((rc = check_launch(c, "k_kmer_count_kept"))) return done(rc);
    unsigned long long tot[2] = {0, 0};
    if((rc = dl(c, tot, dTot, 2))) return done(rc);
    HIP_TRY_F(c, hipStreamSynchronize(c->active), done);
    if(tot[0] == 0) return done(HLALA_OK);
    hlala_ctx::KeptReads kr; kr.n = (int)tot[0];
    void* p = nullptr;
    if((rc = pool_malloc(c, &p, (size_t)tot[1] + 64))) return done(rc); kr.store = (uint8_t*)p;
    if((rc = pool_malloc(c, &p, (size_t)tot[0] * sizeof(long long)))) { pool_release(c, kr.store); return done(rc); } kr.start = (long long*)p;
    if((rc = pool_malloc(c, &p, (size_t)tot[0] * sizeof(int)))) { pool_release(c, kr.store); pool_release(c, kr.start); return done(rc); } kr.length = (int*)p;
    // (the chunk joins the context's list only once it is filled: a failed launch or copy must not leave an entry with undefined start[] / length[] behind,
    //  which hlala_kmer_presence_kept would index the store with)
    auto drop = [&](int r_) { (void)hipStreamSynchronize(c->active); pool_release(c, kr.store); pool_release(c, kr.start); pool_release(c, kr.length); return done(r_); };
    const unsigned grid = (unsigned)std::min<long long>((long long)nReads, (long long)c->stitch_grid);
    hipLaunchKernelGGL(k_kmer_keep, dim3(grid), dim3(64), 0, c->active, b->dB, (const uint8_t*)dMask, dTot + 2, kr.start, kr.length, kr.store);
    if((rc = check_launch(c, "k_kmer_keep"))) return drop(rc);
    if(hipStreamSynchronize(c->active) != hipSuccess) { c->err = "hlala_kmer_keep_reads: the device reported an error"; return drop(HLALA_E_DEVICE); }      // (the mask and the cursors go back to the pool)
    c->kept.push_back(kr);                                                          // (owned by the context from here on)
    if(n_reads_kept) *n_reads_kept = (int64_t)tot[0];
    return done(HLALA_OK);
}

extern "C" int hlala_kmer_presence_kept(hlala_ctx* c, int32_t k, int32_t n_queries, const char* queries, uint8_t* present)
{
    DEV_GUARD(c);
    ReaderScope rscope_(c, nullptr); if(rscope_.rc) return rscope_.rc;
    if(!c || n_queries < 0 || (n_queries > 0 && (!queries || !present))) return HLALA_E_ARG;
    std::vector<u64> canon, uniq;
    int rc = kmer_queries(c, "hlala_kmer_presence_kept", k, n_queries, queries, canon, uniq); if(rc) return rc;
    if(n_queries == 0) return HLALA_OK;
    memset(present, 0, (size_t)n_queries);
    if(uniq.empty() || c->kept.empty()) return HLALA_OK;
    std::vector<void*> tmp;
    auto done = [&](int r_) { for(void* p : tmp) pool_release(c, p); return r_; };
    u64* dQ = nullptr; uint8_t* dP = nullptr;
    if((rc = dev_upload(c, tmp, uniq.data(), uniq.size(), &dQ)) || (rc = dev_alloc(c, tmp, uniq.size(), &dP, true))) return done(rc);
    for(const hlala_ctx::KeptReads& kr : c->kept) {
        const unsigned grid = (unsigned)std::min<long long>((long long)kr.n, (long long)c->stitch_grid);
        hipLaunchKernelGGL(k_kmer_presence_kept, dim3(grid), dim3(64), 0, c->active, (const long long*)kr.start, (const int*)kr.length, (const uint8_t*)kr.store, kr.n, (int)k, (int)uniq.size(), (const u64*)dQ, dP);
        if((rc = check_launch(c, "k_kmer_presence_kept"))) return done(rc);
    }
    std::vector<uint8_t> hp(uniq.size());
    if((rc = dl(c, hp.data(), dP, uniq.size()))) return done(rc);
    HIP_TRY_F(c, hipStreamSynchronize(c->active), done);
    for(int i = 0; i < n_queries; i++) if(canon[i] != ~0ull) present[i] = hp[(size_t)(std::lower_bound(uniq.begin(), uniq.end(), canon[i]) - uniq.begin())];
    return done(HLALA_OK);
}

// ---- page-locked host memory (include/hlala_gpu.h)
extern "C" void* hlala_pinned_alloc(size_t bytes)
{
    void* p = nullptr;
    if(hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    return p;
}
extern "C" void hlala_pinned_free(void* p) { if(p) (void)hipHostFree(p); }
extern "C" int hlala_host_register(void* p, size_t bytes)
{
    if(!p || !bytes) return HLALA_E_ARG;
    if(hipHostRegister(p, bytes, hipHostRegisterDefault) != hipSuccess) { (void)hipGetLastError(); return HLALA_E_DEVICE; }
    return HLALA_OK;
}
extern "C" int hlala_host_unregister(void* p)
{
    if(!p) return HLALA_E_ARG;
    if(hipHostUnregister(p) != hipSuccess) { (void)hipGetLastError(); return HLALA_E_DEVICE; }
    return HLALA_OK;
}
static void seed_batch_unpin(hlala_seed_batch* S)
{
    if(hlala_host::seed_batch_pin_lazy(S)) {
        std::lock_guard<std::mutex> g(hlala_host::seed_batch_pin_mutex(S));
        for(auto& r : hlala_host::seed_batch_pin_regions(S)) { if(hipHostUnregister(r.first) != hipSuccess) (void)hipGetLastError(); }
        hlala_host::seed_batch_pin_regions(S).clear(); hlala_host::seed_batch_pin_cursor(S).clear();
        hlala_host::seed_batch_pin_lazy(S) = false;
    } else {
        std::vector<std::pair<void*, size_t>> arr; hlala_host::seed_batch_bulk_arrays(S, arr);
        for(auto& a : arr) { if(hipHostUnregister(a.first) != hipSuccess) (void)hipGetLastError(); }
    }
    hlala_host::seed_batch_pinned_flag(S) = false;
}
// window by window (pin = 2): what the units before unit_end occupy of every bulk array, rounded up to the next multiple of 64 MB, beyond what is locked already.  Called by
// hlala_seed_batch_window after it has filled the window; a refusal only costs speed (the upload of that window goes through the driver's staging buffer).
static void seed_batch_pin_upto(hlala_seed_batch* S, int64_t unit_end)
{
    std::lock_guard<std::mutex> g(hlala_host::seed_batch_pin_mutex(S));
    std::vector<std::pair<void*, size_t>> arr; std::vector<size_t> upto;
    hlala_host::seed_batch_bulk_arrays(S, arr, unit_end, &upto);
    std::vector<size_t>& cur = hlala_host::seed_batch_pin_cursor(S);
    if(cur.size() != arr.size()) cur.assign(arr.size(), 0);
    // pieces begin and end on multiples of PIN_GRANULE bytes of ADDRESS (the array's own start and end excepted): dev_upload cuts its copies there
    for(size_t i = 0; i < arr.size(); i++) {
        const uintptr_t base = (uintptr_t)arr[i].first;
        const size_t target = std::min(arr[i].second, (size_t)(((base + upto[i] + PIN_GRANULE - 1) & ~(uintptr_t)(PIN_GRANULE - 1)) - base));
        if(target <= cur[i]) continue;
        void* p = (char*)arr[i].first + cur[i];
        if(hipHostRegister(p, target - cur[i], hipHostRegisterDefault) == hipSuccess) hlala_host::seed_batch_pin_regions(S).emplace_back(p, target - cur[i]);
        else (void)hipGetLastError();
        cur[i] = target;
    }
}
extern "C" int hlala_seed_batch_pin(hlala_seed_batch* S, int pin)
{
    if(!S || pin < 0 || pin > 2) return HLALA_E_ARG;
    bool& flag = hlala_host::seed_batch_pinned_flag(S);
    const int now = !flag ? 0 : hlala_host::seed_batch_pin_lazy(S) ? 2 : 1;
    if(pin == now) return HLALA_OK;
    if(flag) seed_batch_unpin(S);
    if(!pin) return HLALA_OK;
    hlala_host::g_seed_batch_unpin = seed_batch_unpin;
    if(pin == 2) { hlala_host::g_seed_batch_pin_upto = seed_batch_pin_upto; hlala_host::seed_batch_pin_lazy(S) = true; flag = true; return HLALA_OK; }
    std::vector<std::pair<void*, size_t>> arr; hlala_host::seed_batch_bulk_arrays(S, arr);
    for(size_t i = 0; i < arr.size(); i++)
        if(hipHostRegister(arr[i].first, arr[i].second, hipHostRegisterDefault) != hipSuccess) {
            (void)hipGetLastError();
            for(size_t k = 0; k < i; k++) (void)hipHostUnregister(arr[k].first);
            return HLALA_E_DEVICE;
        }
    flag = true;
    return HLALA_OK;
}

extern "C" int hlala_abi_sizeof(const char* name)
{
    if(!name) return -1;
    const std::string n(name);
#define SZ(t) if(n == #t) return (int)sizeof(t);
    SZ(hlala_graph_desc) SZ(hlala_contigs_desc) SZ(hlala_params) SZ(hlala_graph_info) SZ(hlala_batch_in) SZ(hlala_seeds_in)
    SZ(hlala_chains_out) SZ(hlala_pairs_out) SZ(hlala_batch_stats) SZ(hlala_exon_in) SZ(hlala_call_out) SZ(hlala_locus_desc) SZ(hlala_exon_positions_out) SZ(hlala_filter_params) SZ(hlala_filter_stats) SZ(hlala_insert_size_out) SZ(hlala_locus_info) SZ(hlala_locus_report_in) SZ(hlala_locus_report_out) SZ(hlala_unit_stats_out) SZ(hlala_pairs_packed_out)
#undef SZ
    return -1;
}

extern "C" int hlala_abi_version(void) { return HLALA_ABI_VERSION; }
extern "C" int hlala_build_flags(void)
{
    int f = 0;
#ifdef HLALA_DP_AGENT_RELEASE
    f |= HLALA_BUILD_AGENT_RELEASE;
#endif
    return f;
}

extern "C" int hlala_kat_exp(hlala_ctx* c, int n, const double* x, double* y)
{
    DEV_GUARD(c);
    if(!c || n < 0 || !x || !y) return HLALA_E_ARG;
    if(!n) return HLALA_OK;
    double *dx = nullptr, *dy = nullptr; std::vector<void*> tmp;
    int rc = dev_upload(c, tmp, x, (size_t)n, &dx); if(rc) return rc;
    rc = dev_alloc(c, tmp, (size_t)n, &dy); if(rc) return rc;
    hipLaunchKernelGGL(k_kat_exp, dim3((n + 255) / 256), dim3(256), 0, c->active, n, (const double*)dx, dy);
    HIP_TRY(c, hipMemcpyAsync(y, dy, (size_t)n * 8, hipMemcpyDeviceToHost, c->active));
    HIP_TRY(c, hipStreamSynchronize(c->active));
    for(void* p : tmp) pool_release(c, p);
    return HLALA_OK;
}

extern "C" int hlala_kat_rand_r(hlala_ctx* c, int n, uint32_t* seeds_inout, int32_t* values_out)
{
    DEV_GUARD(c);
    if(!c || n < 0 || !seeds_inout || !values_out) return HLALA_E_ARG;
    if(!n) return HLALA_OK;
    u32* ds = nullptr; int* dv = nullptr; std::vector<void*> tmp;
    int rc = dev_upload(c, tmp, seeds_inout, (size_t)n, &ds); if(rc) return rc;
    rc = dev_alloc(c, tmp, (size_t)n, &dv); if(rc) return rc;
    hipLaunchKernelGGL(k_kat_rand, dim3((n + 255) / 256), dim3(256), 0, c->active, n, ds, dv);
    HIP_TRY(c, hipMemcpyAsync(seeds_inout, ds, (size_t)n * 4, hipMemcpyDeviceToHost, c->active));
    HIP_TRY(c, hipMemcpyAsync(values_out, dv, (size_t)n * 4, hipMemcpyDeviceToHost, c->active));
    HIP_TRY(c, hipStreamSynchronize(c->active));
    for(void* p : tmp) pool_release(c, p);
    return HLALA_OK;
}
