// host_bam.cpp -- BAM reader + seed extraction: the records of a bwa-mem BAM that fall into the PRG intervals become the
// pairs / chains of an hlala_batch_in.  Sized for a whole sample (BASELINE config 3: ~10 M pairs, ~60 M records, > 2^31 read bases):
// 64-bit offsets, every phase parallel.
//
// Reference: processBAM::extractSeeds2 (mapper/processBAM.cpp:703-864), reads::protoSeeds::takeAlignment / isComplete
// (mapper/reads/protoSeeds.cpp:23-36, 371-380), sortChainsInSeeds (:1945-1967), getAlignmentScore (:4314-4334: the AS tag), the order of
// completeProtoSeeds (std::map over read names, :2024-2039).  BamTools (un-vendored, "tested with 2.5.1", makefile:3-12) is replaced by
// a direct reading of the BAM format (SAM/BAM specification v1: BGZF blocks = gzip members with a BC extra field, little-endian records).
//   * a record is used if it is mapped, (long-read mode: primary,) its reference carries intervals, it has CIGAR operations, and both its
//     start and its end (Position + reference-consuming length - 1 = GetEndPosition(false, true)) lie inside an interval (:763-768);
//   * positions are re-based to the interval start (the reference passes the interval start as reference2level_offset_0based and indexes
//     the translation with position - offset; here chain_pos = position - start and chain_offset = 0, the contig is the interval);
//   * std::sort + std::reverse on the alignment scores is the reference's own call (:1952-1961): equal scores keep the library's order,
//     so the alignments of a mate are collected in FILE order first (every kept record carries its sequence number).
//
// Phases (wall clock of each in hlala_seed_batch_timing):
//   index   the file is mapped and the BGZF block headers are walked (block boundaries are only known sequentially; ~16 bytes per 64 KB)
//   inflate blocks are independent gzip members: a segment of consecutive blocks is inflated by all threads into one buffer
//   parse   record boundaries of the segment are found by hopping over the 4-byte length fields, then ranges of records are parsed in
//           parallel: filters, CIGAR, AS tag, name hash; bases / qualities are unpacked for primary records only (the only ones the path
//           reads, processBAM.cpp:3142-3145); kept records go to one of 256 partitions by name hash
//   group   every partition is grouped by read name on its own (sort by hash, names compared within equal hashes)
//   sort    the complete units of all partitions are put into read-name order by a parallel sample sort (byte-wise comparison then
//           length = std::string's operator<, the order of the reference's std::map)
//   layout  prefix sums over the units, then the output arrays are filled in parallel
#include <sys/resource.h>
#include <sys/syscall.h>
#include <unistd.h>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>
#include <dlfcn.h>

#include "../../include/hlala_gpu.h"
#include "host_internal.h"

namespace {

thread_local std::string g_bam_error;

typedef std::chrono::steady_clock Clock;
double since(Clock::time_point t0) { return std::chrono::duration<double>(Clock::now() - t0).count(); }

// HLALA_BAM_NICE=n: the decoder's worker threads lower their own priority by n (setpriority on the thread).  A host program that decodes one sample while it aligns another
// keeps the threads that feed the GPU ahead of the decoder's -- there are twice as many of those as the control group grants CPUs (experiment: tools/gpu_r6_e2e2.sh).
static void worker_nice()
{
    static const int n = []() { const char* e = getenv("HLALA_BAM_NICE"); return e ? atoi(e) : 0; }();
    if(n > 0) (void)setpriority(PRIO_PROCESS, (id_t)syscall(SYS_gettid), n);
}

// run fn(task) for task in [0, n) on up to T threads (dynamic distribution); the first exception is rethrown on the caller's thread
template <class F>
void parallel_for(int64_t n, int T, F fn)
{
    if(n <= 0) return;
    if(T > n) T = (int)n;
    if(T <= 1) { for(int64_t i = 0; i < n; i++) fn(i, 0); return; }
    std::atomic<int64_t> next(0);
    std::exception_ptr err; std::mutex em;
    std::vector<std::thread> th;
    try {
        for(int t = 0; t < T; t++) th.emplace_back([&, t]() {
            try { worker_nice(); for(;;) { int64_t i = next.fetch_add(1); if(i >= n) break; fn(i, t); } }
            catch(...) { std::lock_guard<std::mutex> g(em); if(!err) err = std::current_exception(); next.store(n); }
        });
    } catch(...) { next.store(n); for(auto& x : th) x.join(); throw; }       // (a thread that cannot be started: the ones running are joined, not destroyed joinable)
    for(auto& x : th) x.join();
    if(err) std::rethrow_exception(err);
}

// fn(i, t) for i = 0 .. n-1 on T threads, the items handed out in ascending order, and beside() on the calling thread while they run.  beside(stop) follows
// the items (it watches flags fn sets) and must return when `stop` turns true: a worker has failed and the remaining items will never be done.
template <class F, class M>
void parallel_for_beside(int64_t n, int T, F fn, M beside)
{
    std::atomic<bool> stop(false);
    if(n <= 0 || T <= 1) { for(int64_t i = 0; i < n; i++) fn(i, 0); beside(stop); return; }
    if(T > n) T = (int)n;
    std::atomic<int64_t> next(0);
    std::exception_ptr err, errMain; std::mutex em;
    std::vector<std::thread> th;
    try {
        for(int t = 0; t < T; t++) th.emplace_back([&, t]() {
            try { worker_nice(); for(;;) { int64_t i = next.fetch_add(1); if(i >= n) break; fn(i, t); } }
            catch(...) { std::lock_guard<std::mutex> g(em); if(!err) err = std::current_exception(); next.store(n); stop.store(true); }
        });
    } catch(...) { next.store(n); stop.store(true); for(auto& x : th) x.join(); throw; }
    try { beside(stop); } catch(...) { errMain = std::current_exception(); next.store(n); }
    for(auto& x : th) x.join();
    if(err) std::rethrow_exception(err);
    if(errMain) std::rethrow_exception(errMain);
}

struct Fail : std::runtime_error { using std::runtime_error::runtime_error; };

// Raw DEFLATE of one BGZF block.  zlib's inflate is what the image links against; libdeflate (whole-buffer decoder, about twice as fast on BGZF blocks) is used
// instead when the machine has its shared library -- looked up at run time, its three entry points declared here (the image ships the library without a header).
// Both produce the same bytes; HLALA_BAM_ZLIB=1 keeps zlib (tests/test_bam.py decodes one file both ways).
struct Inflater {
    typedef void* (*AllocFn)(void); typedef int (*RunFn)(void*, const void*, size_t, void*, size_t, size_t*); typedef void (*FreeFn)(void*);
    static AllocFn ld_alloc; static RunFn ld_run; static FreeFn ld_free; static std::once_flag once;
    static void look()
    {
        if(getenv("HLALA_BAM_ZLIB")) return;
        void* h = dlopen("libdeflate.so.0", RTLD_NOW | RTLD_LOCAL);
        if(!h) h = dlopen("libdeflate.so", RTLD_NOW | RTLD_LOCAL);
        if(!h) return;
        AllocFn a = (AllocFn)dlsym(h, "libdeflate_alloc_decompressor"); RunFn r = (RunFn)dlsym(h, "libdeflate_deflate_decompress"); FreeFn f = (FreeFn)dlsym(h, "libdeflate_free_decompressor");
        if(a && r && f) { ld_alloc = a; ld_run = r; ld_free = f; }
    }
    void* d = nullptr;
    Inflater() { std::call_once(once, look); if(ld_alloc) d = ld_alloc(); }
    ~Inflater() { if(d) ld_free(d); }
    Inflater(const Inflater&) = delete; Inflater& operator=(const Inflater&) = delete;
    static const char* engine() { std::call_once(once, look); return ld_alloc ? "libdeflate" : "zlib"; }
    void run(const uint8_t* in, size_t nin, uint8_t* out, size_t nout)
    {
        if(d) { size_t got = 0; if(ld_run(d, in, nin, out, nout, &got) != 0 || got != nout) throw Fail("BGZF inflate failed"); return; }
        z_stream zs; memset(&zs, 0, sizeof(zs));
        if(inflateInit2(&zs, -15) != Z_OK) throw Fail("inflateInit2 failed");
        zs.next_in = (Bytef*)in; zs.avail_in = (uInt)nin; zs.next_out = out; zs.avail_out = (uInt)nout;
        const int rc = inflate(&zs, Z_FINISH); const uLong got = zs.total_out; inflateEnd(&zs);
        if(rc != Z_STREAM_END || got != nout) throw Fail("BGZF inflate failed");
    }
};
Inflater::AllocFn Inflater::ld_alloc = nullptr; Inflater::RunFn Inflater::ld_run = nullptr; Inflater::FreeFn Inflater::ld_free = nullptr; std::once_flag Inflater::once;

uint32_t rd32(const uint8_t* p) { return p[0] | (p[1] << 8) | (p[2] << 16) | ((uint32_t)p[3] << 24); }

// ---- the mapped file (owning: released on every path out of the extraction)
struct MappedFile {
    int fd = -1; const uint8_t* p = nullptr; size_t n = 0;
    ~MappedFile() { if(p && n) munmap((void*)p, n); if(fd >= 0) close(fd); }
};

struct Block { size_t coff; uint32_t clen, isize; uint64_t uoff; };      // compressed payload at coff (clen bytes), uncompressed isize bytes at uoff of the stream

// one kept alignment (per interval it falls into, as in the reference's loop :746-830)
struct Rec {
    uint64_t hash, order;          // name hash; sequence number in the file (x 256 + interval rank): the order within a mate before sortChainsInSeeds
    const uint8_t* rec;            // the BAM record (after its length field) in the inflated data, which is kept until the sample is laid out: name, CIGAR, packed bases
                                   // and qualities are read from there -- nothing is copied at parse time
    int32_t contig, pos, as, l_seq;        // l_seq: 0 for non-primary records (their bases are never read)
    uint16_t n_cigar, nameLen; uint8_t which, flags, l_read_name, pad1;       // flags: 1 reverse, 2 primary
    const char* name() const { return (const char*)rec + 32; }
    const uint8_t* cigar() const { return rec + 32 + l_read_name; }                       // 4 bytes per operation, little-endian, unaligned
    const uint8_t* seq4() const { return cigar() + 4 * (size_t)n_cigar; }                 // 4-bit bases, two per byte
    const uint8_t* qual(int32_t lseq) const { return seq4() + ((size_t)lseq + 1) / 2; }   // Phred values
};
// large allocations: page aligned, huge pages where the kernel grants them (first-touch page faults of multi-GB arrays otherwise cost more than the
// copies that fill them), never cleared
uint8_t* big_alloc(size_t bytes)
{
    constexpr size_t HUGE = (size_t)2 << 20;
    void* p = nullptr;
    if(bytes >= HUGE) {
        if(posix_memalign(&p, HUGE, (bytes + HUGE - 1) & ~(HUGE - 1)) != 0) p = nullptr;
        if(p) (void)madvise(p, (bytes + HUGE - 1) & ~(HUGE - 1), MADV_HUGEPAGE);
    } else p = malloc(bytes ? bytes : 1);
    if(!p) throw std::bad_alloc();
    return (uint8_t*)p;
}
struct BigFree { void operator()(uint8_t* p) const { free(p); } };

// what one decoding thread collects: its kept records by name partition
struct Arena { std::vector<std::vector<Rec>> part; int64_t examined = 0; };

// a big array of the result: allocated without being cleared (the threads that fill it are the first to touch its pages)
template <class T>
struct Buf {
    T* p = nullptr; size_t n = 0;
    Buf() {}
    ~Buf() { free(p); }
    Buf(const Buf&) = delete; Buf& operator=(const Buf&) = delete;
    void alloc(size_t k) { free(p); p = nullptr; n = 0; if(k) { p = (T*)big_alloc(k * sizeof(T)); n = k; } }
    T* data() { return p; } const T* data() const { return p; } size_t size() const { return n; }
    T& operator[](size_t i) { return p[i]; } const T& operator[](size_t i) const { return p[i]; }
};

constexpr int NPART = 256;

uint64_t hash_name(const uint8_t* s, size_t n)
{
    uint64_t h = 0xcbf29ce484222325ull;
    for(size_t i = 0; i < n; i++) { h ^= s[i]; h *= 0x100000001b3ull; }
    h ^= h >> 29; h *= 0xbf58476d1ce4e5b9ull; h ^= h >> 32;
    return h;
}

struct Unit { const char* name; uint32_t nameLen; uint32_t part; uint32_t first, count; };      // recs [first, first + count) of the partition's sorted record list

// The decoder's working memory: the inflated rounds (the records the Recs point into), the kept records by name partition, the units in name order.  It outlives
// hlala_bam_extract_seeds_mt: the arrays of the result are FILLED per window, when a window is first asked for (fill state below), and released when the last
// unit has been filled (or with the seed batch).
struct Work { std::vector<Arena> arenas; std::vector<std::vector<Rec>> precs; std::vector<Unit> units; Buf<int64_t> cigCount;
              std::vector<std::unique_ptr<uint8_t, BigFree>> inflated;
              // ---- fill state: the units are filled in chunks of UCH (chunk c = units [c * UCH, (c + 1) * UCH)), `done` per chunk, under the seed batch's fill_mu
              int nm = 2, T = 1; int64_t UCH = 1, nUChunks = 0, remaining = 0; std::vector<uint8_t> done; };

bool name_less(const Unit& a, const Unit& b)
{
    const size_t n = a.nameLen < b.nameLen ? a.nameLen : b.nameLen;
    const int c = memcmp(a.name, b.name, n);
    return c != 0 ? c < 0 : a.nameLen < b.nameLen;
}

}  // namespace

struct hlala_seed_batch {
    Buf<int64_t> read_off, chain_off;
    Buf<int64_t> cigar_off;
    Buf<int32_t> read_primary;
    Buf<int32_t> chain_contig, chain_pos, chain_offset, chain_as;
    Buf<uint8_t> read_bases, read_quals, chain_reverse; Buf<uint32_t> cigar;
    Buf<uint8_t> read_bases_packed; bool packed = false;      // HLALA_SEEDS_PACKED: the bases stay 4-bit packed as the BAM records hold them (hlala_batch_in::read_bases_packed), read_bases is empty
    Buf<char> name_chars; Buf<int64_t> name_off;          // names of the units, NUL-terminated
    int64_t n_units = 0; int32_t unpaired = 0; int64_t examined = 0, n_seeds = 0, n_incomplete = 0;
    double seconds[6] = {0, 0, 0, 0, 0, 0}; int32_t threads = 1;
    bool pinned = false;
    // page-locking window by window (hlala_seed_batch_pin(S, 2)): per bulk array the bytes locked so far, and the regions to unlock
    bool pin_lazy = false; std::mutex pin_mu; std::vector<size_t> pin_cursor; std::vector<std::pair<void*, size_t>> pin_regions;
    // the decoder's working memory while units remain to be filled (hlala_seed_batch_window fills what it hands out); null once every unit is filled
    Work* work = nullptr; double fill_seconds = 0; std::mutex fill_mu;       // (the mutex lives here, not in the working memory it outlives)
    ~hlala_seed_batch() { if(work) { Work* w = work; work = nullptr; try { std::thread([w]() { delete w; }).detach(); } catch(...) { delete w; } } }
};

namespace {
// sortChainsInSeeds (:1952-1961) on the mate's alignments in file order; returns the position of the first primary (read*_getPrimaryAlignmentI)
size_t sorted_mate(const std::vector<std::vector<Rec>>& precs, const Unit& u, int m, std::vector<uint32_t>& idx)
{
    const std::vector<Rec>& R = precs[u.part];
    idx.clear();
    for(uint32_t k = u.first; k < u.first + u.count; k++) if(R[k].which == m) idx.push_back(k);
    std::sort(idx.begin(), idx.end(), [&](uint32_t a, uint32_t b) { return R[a].as < R[b].as; });
    std::reverse(idx.begin(), idx.end());
    for(size_t i = 0; i < idx.size(); i++) if(R[idx[i]].flags & 2) return i;
    return idx.size();
}

// names, bases, qualities, alignments and CIGARs of the units [a, z) into the arrays of the result (offsets and sizes are in place since the decode)
void fill_units(hlala_seed_batch* S, const Work& W, const size_t a, const size_t z, std::vector<uint32_t>& idx)
{
    static const char SEQ16[] = "=ACMGRSVTWYHKDBN";
    const std::vector<Unit>& units = W.units; const std::vector<std::vector<Rec>>& precs = W.precs; const Buf<int64_t>& cigCount = W.cigCount; const int nm = W.nm;
    for(size_t ui = a; ui < z; ui++) {
        // The fill gathers names, CIGARs and packed bases from records scattered over the whole inflated file (name order against coordinate order: a cache
        // and TLB miss per record, 6.8 us per read and thread on a 128-thread host).  Two steps ahead of the work: the descriptors of the unit sixteen
        // ahead, and -- through the descriptors requested eight units ago -- the record bytes of the unit eight ahead (CIGAR; bases and qualities of a primary).
        if(ui + 16 < z) { const Unit& uf = units[ui + 16]; const Rec* rf = precs[uf.part].data() + uf.first; for(uint32_t k = 0; k < uf.count; k++) __builtin_prefetch(rf + k); }
        if(ui + 8 < z) {
            const Unit& un = units[ui + 8]; const Rec* rn = precs[un.part].data() + un.first;
            __builtin_prefetch(un.name);
            for(uint32_t k = 0; k < un.count; k++) {
                const uint8_t* cg = rn[k].cigar(); __builtin_prefetch(cg);
                if(rn[k].l_seq > 0) { const uint8_t* sq = rn[k].seq4(); const size_t nb = ((size_t)rn[k].l_seq + 1) / 2 + (size_t)rn[k].l_seq; for(size_t o = 0; o < nb + 63; o += 64) __builtin_prefetch(sq + o); }
            }
        }
        const Unit& u = units[ui]; const std::vector<Rec>& R = precs[u.part];
        memcpy(S->name_chars.data() + S->name_off[ui], u.name, (size_t)u.nameLen); S->name_chars[(size_t)S->name_off[ui] + u.nameLen] = 0;
        for(int m = 0; m < nm; m++) {
            const size_t prim = sorted_mate(precs, u, m, idx);
            const size_t r = ui * (size_t)nm + (size_t)m;
            const Rec& pa = R[idx[prim]];
            {   // QueryBases / Qualities of the primary (BuildCharData: 4-bit codes -> characters, Phred + 33), alignment orientation (:3142-3145)
                const int32_t ls = pa.l_seq; const uint8_t* s4 = pa.seq4(); const uint8_t* ql = pa.qual(ls);
                uint8_t* qs = S->read_quals.data() + S->read_off[r];
                if(S->packed) memcpy(S->read_bases_packed.data() + ((S->read_off[r] + (int64_t)r + 1) >> 1), s4, ((size_t)ls + 1) / 2);       // every read on a byte of its own: include/hlala_gpu.h
                else {
                    uint8_t* bs = S->read_bases.data() + S->read_off[r];
                    for(int32_t i = 0; i + 1 < ls; i += 2) { const unsigned b = s4[(size_t)i / 2]; bs[i] = (uint8_t)SEQ16[b >> 4]; bs[i + 1] = (uint8_t)SEQ16[b & 15]; }
                    if(ls & 1) bs[ls - 1] = (uint8_t)SEQ16[s4[(size_t)(ls - 1) / 2] >> 4];
                }
                for(int32_t i = 0; i < ls; i++) qs[i] = (uint8_t)(ql[i] + 33);
            }
            size_t ch = (size_t)S->chain_off[r]; int64_t cg = cigCount[r];
            S->read_primary[r] = (int32_t)(ch + prim);
            for(uint32_t k : idx) {
                const Rec& x = R[k];
                S->chain_contig[ch] = x.contig; S->chain_pos[ch] = x.pos; S->chain_offset[ch] = 0; S->chain_as[ch] = x.as; S->chain_reverse[ch] = (uint8_t)(x.flags & 1);
                memcpy(S->cigar.data() + cg, x.cigar(), 4 * (size_t)x.n_cigar);               // (BAM is little-endian like every host this library runs on)
                cg += x.n_cigar; S->cigar_off[ch + 1] = cg; ch++;
            }
        }
    }
}

// The units [u0, u1) are filled before this returns (whichever thread asks first fills them, on the decoder's thread count; the others wait).  Once no unit is left
// the working memory is released on a thread of its own (unmapping a dozen GB takes about a second that the caller need not wait for).
void ensure_filled(const hlala_seed_batch* Sc, int64_t u0, int64_t u1)
{
    hlala_seed_batch* S = const_cast<hlala_seed_batch*>(Sc);
    if(u1 <= u0) return;
    std::unique_lock<std::mutex> g(S->fill_mu);
    Work* W = S->work;
    if(!W) return;
    const Clock::time_point t0 = Clock::now();
    std::vector<int64_t> todo;
    for(int64_t c = u0 / W->UCH; c <= (u1 - 1) / W->UCH && c < W->nUChunks; c++) if(!W->done[(size_t)c]) todo.push_back(c);
    if(!todo.empty()) {
        std::vector<std::vector<uint32_t>> order((size_t)W->T);
        const size_t nU = W->units.size();
        parallel_for((int64_t)todo.size(), W->T, [&](int64_t i, int t) {
            const int64_t c = todo[(size_t)i];
            fill_units(S, *W, (size_t)(c * W->UCH), std::min(nU, (size_t)((c + 1) * W->UCH)), order[(size_t)t]);
        });
        for(int64_t c : todo) W->done[(size_t)c] = 1;
        W->remaining -= (int64_t)todo.size();
    }
    const double dt = since(t0);
    S->fill_seconds += dt; S->seconds[5] += dt;
    if(W->remaining <= 0) {
        S->work = nullptr;
        g.unlock();
        try { std::thread([W]() { delete W; }).detach(); } catch(...) { delete W; }
    }
}
}  // namespace

namespace hlala_host {
int host_cpu_budget()
{
    int hw = (int)std::thread::hardware_concurrency(); if(hw < 1) hw = 1;
    double quota = 0;
    if(FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {                              // cgroup v2: "<quota|max> <period>"
        char q[64]; double per = 0;
        if(fscanf(f, "%63s %lf", q, &per) == 2 && strcmp(q, "max") != 0 && per > 0) quota = atof(q) / per;
        fclose(f);
    } else {
        double q = 0, per = 0;
        if(FILE* g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) { if(fscanf(g, "%lf", &q) != 1) q = 0; fclose(g); }
        if(FILE* g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if(fscanf(g, "%lf", &per) != 1) per = 0; fclose(g); }
        if(q > 0 && per > 0) quota = q / per;
    }
    if(quota > 0) { const int c = (int)(quota + 0.999); if(c >= 1 && c < hw) hw = c; }
    return hw;
}
// the bulk arrays of a seed batch (what a batch upload reads most of its bytes from) and, with `upto`, how many bytes of each the units before unit_end occupy
void seed_batch_bulk_arrays(hlala_seed_batch* S, std::vector<std::pair<void*, size_t>>& out, int64_t unit_end, std::vector<size_t>* upto)
{
    out.clear(); if(upto) upto->clear();
    size_t nBases = 0, nPacked = 0, nChains = 0, nCigar = 0;
    if(upto) {
        const size_t r = (size_t)std::max<int64_t>(0, std::min(unit_end, S->n_units)) * (size_t)(S->unpaired ? 1 : 2);
        nBases = (size_t)S->read_off[r]; nPacked = (nBases + r + 1) >> 1; nChains = (size_t)S->chain_off[r]; nCigar = (size_t)S->cigar_off[nChains];
    }
    auto add = [&](void* p, size_t b, size_t u) { if(p && b) { out.emplace_back(p, b); if(upto) upto->push_back(std::min(u, b)); } };
    add(S->read_bases.data(), S->read_bases.size(), nBases); add(S->read_bases_packed.data(), S->read_bases_packed.size(), nPacked); add(S->read_quals.data(), S->read_quals.size(), nBases);
    add(S->chain_contig.data(), S->chain_contig.size() * 4, nChains * 4); add(S->chain_pos.data(), S->chain_pos.size() * 4, nChains * 4); add(S->chain_offset.data(), S->chain_offset.size() * 4, nChains * 4);
    add(S->chain_as.data(), S->chain_as.size() * 4, nChains * 4); add(S->chain_reverse.data(), S->chain_reverse.size(), nChains); add(S->cigar.data(), S->cigar.size() * 4, nCigar * 4);
}
void seed_batch_bulk_arrays(hlala_seed_batch* S, std::vector<std::pair<void*, size_t>>& out) { seed_batch_bulk_arrays(S, out, 0, nullptr); }
bool& seed_batch_pin_lazy(hlala_seed_batch* S) { return S->pin_lazy; }
std::mutex& seed_batch_pin_mutex(hlala_seed_batch* S) { return S->pin_mu; }
std::vector<size_t>& seed_batch_pin_cursor(hlala_seed_batch* S) { return S->pin_cursor; }
std::vector<std::pair<void*, size_t>>& seed_batch_pin_regions(hlala_seed_batch* S) { return S->pin_regions; }
void (*g_seed_batch_pin_upto)(hlala_seed_batch*, int64_t) = nullptr;      // set by the GPU library: page-locks what the units before unit_end occupy and is not locked yet
bool& seed_batch_pinned_flag(hlala_seed_batch* S) { return S->pinned; }
void (*g_seed_batch_unpin)(hlala_seed_batch*) = nullptr;       // set by the GPU library (hlala_seed_batch_pin): a pinned batch is unpinned before it is freed
}  // namespace hlala_host

// (host code: no device involved)
extern "C" int hlala_pack_bases(const uint8_t* read_bases, const int64_t* read_off, int64_t n_reads, uint8_t* packed)
{
    if(!read_bases || !read_off || !packed || n_reads < 0) return HLALA_E_ARG;
    // (a function-local static initialised by a lambda: thread-safe by the language; two callers making the first call at once used to race on a plain flag)
    struct Table { uint8_t v[256]; };
    static const Table table = [] { Table t; memset(t.v, 15, sizeof(t.v)); const char* s16 = "=ACMGRSVTWYHKDBN"; for(int i = 0; i < 16; i++) t.v[(unsigned char)s16[i]] = (uint8_t)i; return t; }();
    const uint8_t* code = table.v;
    for(int64_t r = 0; r < n_reads; r++) {
        const int64_t o = read_off[r], len = read_off[r + 1] - o; uint8_t* dst = packed + ((o + r + 1) >> 1);
        for(int64_t j = 0; j + 1 < len; j += 2) dst[j >> 1] = (uint8_t)((code[read_bases[o + j]] << 4) | code[read_bases[o + j + 1]]);
        if(len & 1) dst[len >> 1] = (uint8_t)(code[read_bases[o + len - 1]] << 4);
    }
    return HLALA_OK;
}
extern "C" const char* hlala_bam_last_error() { return g_bam_error.c_str(); }
extern "C" const char* hlala_bam_inflate_engine() { return Inflater::engine(); }

extern "C" int hlala_bam_extract_seeds(const char* path, int32_t n_intervals, const hlala_bam_interval* iv, int32_t long_read_mode, hlala_seed_batch** out)
{
    return hlala_bam_extract_seeds_mt(path, n_intervals, iv, long_read_mode, 0, out);
}

extern "C" int hlala_bam_extract_seeds_mt(const char* path, int32_t n_intervals, const hlala_bam_interval* iv, int32_t long_read_mode, int32_t n_threads, hlala_seed_batch** out)
{
    return hlala_bam_extract_seeds_opt(path, n_intervals, iv, long_read_mode, n_threads, 0, out);
}

extern "C" int hlala_bam_extract_seeds_opt(const char* path, int32_t n_intervals, const hlala_bam_interval* iv, int32_t long_read_mode, int32_t n_threads, int32_t flags, hlala_seed_batch** out)
try {
    if(!path || !out || n_intervals < 0 || (n_intervals > 0 && !iv) || n_threads < 0) return HLALA_E_ARG;
    *out = nullptr; g_bam_error.clear();
    int T = n_threads;
    // Default: at most 32 threads.  Measured on a 256-thread host (tools/gpu_decode_threads.sh, 8.4 M pairs, 2.6 GB of BAM): 5.2 s on 8 threads, 3.5 on 16, 2.85 on 32,
    // 3.0 on 64, 3.3-3.7 on 128 -- the phases are passes over ~12 GB of inflated records bound by memory latency and by the first touch of their
    // working memory (4 KB pages fault in at 10 GB/s on that host whatever the number of threads); beyond 32 threads they only get in each other's way.
    // It is also the share of such a host that one of eight samples gets (BASELINE config 4).
    if(T == 0) { T = 2 * hlala_host::host_cpu_budget(); if(T > 32) T = 32; }      // (twice the CPUs: the phases wait on memory as much as they compute; beyond 32 threads they only get in each other's way)
    if(T > 1024) T = 1024;
    std::unique_ptr<hlala_seed_batch> S(new hlala_seed_batch());
    S->threads = T; S->unpaired = long_read_mode ? 1 : 0; S->packed = (flags & HLALA_SEEDS_PACKED) != 0;
    auto tPhase = Clock::now();

    // ---------------------------------------------------------------- index: map the file, walk the block headers
    MappedFile mf;
    mf.fd = open(path, O_RDONLY);
    if(mf.fd < 0) throw Fail(std::string("Cannot open BAM file: ") + path);
    { struct stat st; if(fstat(mf.fd, &st) != 0) throw Fail("Cannot stat BAM file"); mf.n = (size_t)st.st_size; }
    if(mf.n == 0) throw Fail("not a BAM file");
    mf.p = (const uint8_t*)mmap(nullptr, mf.n, PROT_READ, MAP_PRIVATE, mf.fd, 0);
    if(mf.p == MAP_FAILED) { mf.p = nullptr; throw Fail("Cannot map BAM file"); }
    (void)madvise((void*)mf.p, mf.n, MADV_SEQUENTIAL);
    std::vector<Block> blocks;
    {
        size_t o = 0; uint64_t u = 0;
        while(o < mf.n) {
            if(mf.n - o < 18) throw Fail("truncated BGZF header");
            const uint8_t* h = mf.p + o;
            if(h[0] != 31 || h[1] != 139 || h[2] != 8 || !(h[3] & 4)) throw Fail("not a BGZF block");
            const size_t xlen = h[10] | (h[11] << 8);
            if(mf.n - o < 12 + xlen) throw Fail("truncated BGZF header");
            // the BC subfield holds the block size - 1; it is the first subfield in every BGZF writer, but walk the extra field anyway
            long bsize = -1;
            for(size_t p = 0; p + 4 <= xlen;) { const uint8_t* e = h + 12 + p; const size_t sl = e[2] | (e[3] << 8); if(e[0] == 66 && e[1] == 67 && sl == 2 && p + 6 <= xlen) bsize = e[4] | (e[5] << 8); p += 4 + sl; }
            if(bsize < 0) throw Fail("BGZF block without BC field");
            if((size_t)bsize + 1 < 12 + xlen + 8) throw Fail("BGZF block size smaller than its own header");
            if(mf.n - o < (size_t)bsize + 1) throw Fail("truncated BGZF block");
            Block b; b.coff = o + 12 + xlen; b.clen = (uint32_t)((size_t)bsize + 1 - 12 - xlen - 8); b.isize = rd32(h + bsize + 1 - 4); b.uoff = u;
            if(b.isize > (1u << 16)) throw Fail("BGZF block with more than 64 KiB of payload");
            if(b.isize) blocks.push_back(b);                              // (empty blocks: the end-of-file marker)
            u += b.isize; o += (size_t)bsize + 1;
        }
    }
    S->seconds[0] = since(tPhase);

    // ---------------------------------------------------------------- inflate + parse, segment by segment
    std::unordered_map<std::string, std::vector<int>> intervalsOfRef;                     // interestingIntervals, processBAM.cpp:1226-1400
    for(int i = 0; i < n_intervals; i++) { if(!iv[i].ref_name || iv[i].stop_0based < iv[i].start_0based) throw Fail("bad interval"); intervalsOfRef[iv[i].ref_name].push_back(i); }
    // the decoder's working memory (a dozen GB for a 10 M-pair sample) is released on a thread of its own after the result is handed over: unmapping it
    // costs about a second that the caller need not wait for
    std::unique_ptr<Work> W(new Work());
    std::vector<Arena>& arenas = W->arenas; arenas.resize((size_t)T);
    for(Arena& a : arenas) a.part.resize(NPART);
    {   // The kept records of a thread go to 256 vectors (one per name partition).  Growing 128 x 256 vectors from nothing made the first round take six times as
        // long as the later ones (reallocation, fresh pages): they are reserved for the share of the file a thread can expect (records of >= 200 bytes, a quarter
        // on top; the memory is only touched when used).
        uint64_t payload = 0; for(const Block& b : blocks) payload += b.isize;
        const size_t each = (size_t)(payload / 200 / ((uint64_t)T * NPART) * 5 / 4) + 32;
        parallel_for((int64_t)T, T, [&](int64_t t, int) { for(std::vector<Rec>& v : arenas[(size_t)t].part) v.reserve(each); });
    }
    std::vector<std::vector<int>> refIntervals;          // per BAM reference id: the intervals it carries
    bool headerDone = false; int32_t n_ref = 0;
    // uncompressed bytes inflated and parsed per round (HLALA_BAM_SEGMENT_BYTES: the tests choose a few blocks per round to exercise records that
    // straddle rounds; the reference dictionary must fit the first round).  1 GiB per round: threads are started and joined twice per round
    size_t SEG_BYTES = (size_t)1 << 30;
    if(const char* e = getenv("HLALA_BAM_SEGMENT_BYTES")) { const long long v = atoll(e); if(v >= 65536) SEG_BYTES = (size_t)v; }
    // a round's buffer = [bytes of the record that straddles the previous round | this round's blocks]; buffers are kept (records point into them)
    const uint8_t* carryFrom = nullptr; size_t carry = 0; uint64_t recSeq = 0;
    double tInflate = 0, tParse = 0;
    std::vector<Inflater> inflaters((size_t)std::max(1, T));
    // (tests: HLALA_BAM_TEST_HASH_BITS=k keeps the partition byte and k low bits of the name hashes -- many names per hash, the path that real samples never take)
    uint64_t hashMask = ~0ull;
    if(const char* e = getenv("HLALA_BAM_TEST_HASH_BITS")) { const int k = atoi(e); if(k >= 0 && k < 56) hashMask = (0xFFull << 56) | ((1ull << k) - 1); }
    for(size_t b0 = 0; b0 < blocks.size();) {
        size_t b1 = b0; size_t segBytes = 0;
        while(b1 < blocks.size() && (segBytes == 0 || segBytes + blocks[b1].isize <= SEG_BYTES)) { segBytes += blocks[b1].isize; b1++; }
        auto t0 = Clock::now();
        W->inflated.emplace_back(big_alloc(carry + segBytes));
        uint8_t* const bufp = W->inflated.back().get(); const size_t bufn = carry + segBytes;
        if(carry) memcpy(bufp, carryFrom, carry);
        const uint64_t u0 = blocks[b0].uoff;
        // The records of a round are found by hopping over their 4-byte length fields, one after the other from the round's first byte: the calling thread does
        // that BESIDE the threads that inflate the round's blocks (handed out in ascending order), right behind the block they have completed last -- on its own
        // after the inflate the hop was half of the parse phase, the one part of it that no thread count shortens.
        const size_t nBlk = b1 - b0;
        std::unique_ptr<std::atomic<uint8_t>[]> blockDone(new std::atomic<uint8_t>[nBlk]);
        for(size_t k = 0; k < nBlk; k++) blockDone[k].store(0, std::memory_order_relaxed);
        const uint8_t* d = bufp; const size_t dn = bufn;
        size_t o = 0;
        const bool lastSegment = b1 == blocks.size();
        std::vector<size_t> recStart;
        recStart.reserve(dn / 200 + 16);
        parallel_for_beside((int64_t)nBlk, T, [&](int64_t k, int t) {
            const Block& b = blocks[b0 + (size_t)k];
            inflaters[(size_t)t].run(mf.p + b.coff, b.clen, bufp + carry + (size_t)(b.uoff - u0), b.isize);
            blockDone[(size_t)k].store(1, std::memory_order_release);
        }, [&](const std::atomic<bool>& stop) {
            size_t ready = carry, kReady = 0;                   // bytes [0, ready) of the buffer are final: the carried bytes + the blocks before kReady
            // true when bytes [o, o + k) are there to be read (waits for the blocks that hold them); false: the round ends before o + k, or a worker failed
            auto avail = [&](size_t k) -> bool {
                if(dn - o < k) return false;
                while(ready < o + k) {
                    if(kReady < nBlk && blockDone[kReady].load(std::memory_order_acquire)) { ready += blocks[b0 + kReady].isize; kReady++; continue; }
                    if(stop.load()) return false;
                    std::this_thread::sleep_for(std::chrono::microseconds(50));      // (asleep, not spinning: under a CPU quota a spinning thread is paid for by the inflate threads)
                }
                return true;
            };
            if(!headerDone) {
                // magic, header text, reference list (needs the whole header inside the first round: 1 GiB holds any reference dictionary)
                auto need = [&](size_t k) { if(!avail(k)) throw Fail("truncated BAM header"); };
                need(4); if(memcmp(d, "BAM\1", 4) != 0) throw Fail("not a BAM file"); o = 4;
                need(4); const int32_t l_text = (int32_t)rd32(d + o); o += 4; if(l_text < 0 || l_text > (1 << 30)) throw Fail("truncated BAM header");
                need((size_t)l_text); o += (size_t)l_text;
                need(4); n_ref = (int32_t)rd32(d + o); o += 4; if(n_ref < 0) throw Fail("truncated BAM header");
                refIntervals.resize((size_t)n_ref);
                for(int i = 0; i < n_ref; i++) {
                    if(!avail(4)) throw Fail("truncated BAM reference list");
                    const int32_t l_name = (int32_t)rd32(d + o); o += 4;
                    if(l_name < 1 || l_name > (1 << 20) || !avail((size_t)l_name + 4)) throw Fail("truncated BAM reference list");
                    const std::string nm((const char*)d + o, strnlen((const char*)d + o, (size_t)l_name));
                    o += (size_t)l_name + 4;
                    auto it = intervalsOfRef.find(nm);
                    if(it != intervalsOfRef.end()) refIntervals[(size_t)i] = it->second;
                }
                headerDone = true;
            }
            // record boundaries: hop over the length fields
            while(avail(4)) {
                const int32_t bs = (int32_t)rd32(d + o);
                if(bs < 32 || bs > (1 << 28)) throw Fail("truncated BAM record");
                if(dn - o - 4 < (size_t)bs) break;
                recStart.push_back(o); o += 4 + (size_t)bs;
            }
        });
        tInflate += since(t0); t0 = Clock::now();
        if(lastSegment && o != dn) throw Fail("truncated BAM record");
        const size_t nRec = recStart.size();
        const int64_t CH = std::max<int64_t>(16, std::min<int64_t>(4096, (int64_t)nRec / ((int64_t)T * 8) + 1)); const int64_t nTasks = ((int64_t)nRec + CH - 1) / CH;      // (long reads: few, large records per round)
        const uint64_t seq0 = recSeq;
        parallel_for(nTasks, T, [&](int64_t task, int t) {
            Arena& A = arenas[(size_t)t];
            const size_t r0 = (size_t)(task * CH), r1 = std::min(nRec, r0 + (size_t)CH);
            for(size_t ri = r0; ri < r1; ri++) {
                const uint8_t* rec = d + recStart[ri] + 4; const size_t rn = (size_t)rd32(d + recStart[ri]);
                const int32_t refID = (int32_t)rd32(rec), position = (int32_t)rd32(rec + 4);
                const unsigned l_read_name = rec[8]; const unsigned n_cigar = rec[12] | (rec[13] << 8); const unsigned flag = rec[14] | (rec[15] << 8);
                const int32_t l_seq = (int32_t)rd32(rec + 16);
                if(l_seq < 0) throw Fail("corrupt BAM record");
                const size_t oName = 32, oCigar = oName + l_read_name, oSeq = oCigar + 4 * (size_t)n_cigar, oQual = oSeq + ((size_t)l_seq + 1) / 2, oTags = oQual + (size_t)l_seq;
                if(oTags > rn || l_read_name < 1) throw Fail("corrupt BAM record");
                if(flag & 4) continue;                                                             // ! IsMapped(), :727
                if(long_read_mode && (flag & 256)) continue;                                       // ! IsPrimaryAlignment(), :732-738 (BamTools 2.5.1: !(AlignmentFlag & 0x100))
                if(refID < 0 || refID >= n_ref) continue;
                const std::vector<int>& ivs = refIntervals[(size_t)refID];
                if(ivs.empty()) continue;                                                          // :744
                int refLen = -1; int as = 0; bool haveAS = false, parsed = false;
                uint64_t hash = 0; size_t nameLen = 0;
                int rank = 0;
                for(int ii : ivs) {
                    A.examined++;                                                                  // :757 (per interval, as in the reference)
                    if(n_cigar == 0) continue;                                                     // :759-763
                    if(refLen < 0) { refLen = 0; for(unsigned k = 0; k < n_cigar; k++) { const uint32_t c = rd32(rec + oCigar + 4 * k); const unsigned op = c & 15u; if(op == 0 || op == 2 || op == 3 || op == 7 || op == 8) refLen += (int)(c >> 4); } }
                    const int start = position, stop = position + refLen - 1;                      // GetEndPosition(false, true), :766
                    if(!((start >= iv[ii].start_0based && start <= iv[ii].stop_0based) && (stop >= iv[ii].start_0based && stop <= iv[ii].stop_0based))) continue;
                    if(!long_read_mode && !(flag & 1)) throw Fail("unpaired record in a paired-end BAM (assert(currentAlignment.IsPaired()), processBAM.cpp:783)");
                    const bool primary = !(flag & 256);                                            // BamTools IsPrimaryAlignment: !(flag & 0x100)
                    if(!parsed) {
                        parsed = true;
                        // the AS tag (getAlignmentScore, :4314-4334): any integer type
                        for(size_t p = oTags; p + 3 <= rn;) {
                            const char t0 = (char)rec[p], t1 = (char)rec[p + 1], ty = (char)rec[p + 2]; p += 3;
                            size_t sz = 0; long long v = 0; bool isInt = true;
                            switch(ty) {
                                case 'c': sz = 1; if(p + 1 <= rn) v = (int8_t)rec[p]; break;
                                case 'C': sz = 1; if(p + 1 <= rn) v = rec[p]; break;
                                case 's': sz = 2; if(p + 2 <= rn) v = (int16_t)(rec[p] | (rec[p + 1] << 8)); break;
                                case 'S': sz = 2; if(p + 2 <= rn) v = (uint16_t)(rec[p] | (rec[p + 1] << 8)); break;
                                case 'i': sz = 4; if(p + 4 <= rn) v = (int32_t)rd32(rec + p); break;
                                case 'I': sz = 4; if(p + 4 <= rn) v = rd32(rec + p); break;
                                case 'A': sz = 1; isInt = false; break;
                                case 'f': sz = 4; isInt = false; break;
                                case 'Z': case 'H': { isInt = false; size_t q = p; while(q < rn && rec[q]) q++; sz = q - p + 1; break; }
                                case 'B': { isInt = false; if(p + 5 > rn) throw Fail("corrupt BAM tag"); const char et = (char)rec[p]; const uint32_t cnt = rd32(rec + p + 1);
                                            const size_t es = (et == 'c' || et == 'C') ? 1 : (et == 's' || et == 'S') ? 2 : 4; sz = 5 + es * (size_t)cnt; break; }
                                default: throw Fail("unknown BAM tag type");
                            }
                            if(p + sz > rn) throw Fail("corrupt BAM tag");
                            if(t0 == 'A' && t1 == 'S' && isInt) { as = (int)v; haveAS = true; break; }      // (the first one, as BamTools' GetTag; what follows -- XS, SA, XA: hundreds of bytes with bwa -a -- is not walked)
                            p += sz;
                        }
                        if(!haveAS) throw Fail("Can't get AS tag!");                               // assert(1 == 0), :4330-4332
                        nameLen = strnlen((const char*)rec + oName, l_read_name);
                        hash = hash_name(rec + oName, nameLen) & hashMask;
                    }
                    Rec r; r.hash = hash; r.order = ((seq0 + ri) << 8) | (uint64_t)(rank < 255 ? rank : 255); rank++;
                    r.rec = rec; r.contig = iv[ii].contig; r.pos = position - iv[ii].start_0based; r.as = as; r.l_seq = primary ? l_seq : 0;
                    r.n_cigar = (uint16_t)n_cigar; r.nameLen = (uint16_t)nameLen; r.which = (uint8_t)(long_read_mode ? 0 : ((flag & 64) ? 0 : 1));      // IsFirstMate() ? 1 : 2; long reads: 1 (:814-818)
                    r.flags = (uint8_t)(((flag & 16) ? 1 : 0) | (primary ? 2 : 0)); r.l_read_name = (uint8_t)l_read_name; r.pad1 = 0;
                    A.part[(size_t)(hash >> 56)].push_back(r);
                }
            }
        });
        if(getenv("HLALA_BAM_DEBUG")) fprintf(stderr, "bam-debug: round of %zu blocks, %zu records: inflate (%s; the hop beside it) + parse so far %.3f + %.3f s\n", b1 - b0, nRec, Inflater::engine(), tInflate, tParse + since(t0));
        recSeq += nRec;
        if(recSeq >= (1ull << 55)) throw Fail("more BAM records than the sequence numbers hold");
        // bytes of a record that continues in the next segment move to the front
        carry = dn - o; carryFrom = d + o;
        tParse += since(t0);
        b0 = b1;
    }
    if(!headerDone) throw Fail("not a BAM file");
    S->seconds[1] = tInflate; S->seconds[2] = tParse;
    for(const Arena& a : arenas) S->examined += a.examined;

    // ---------------------------------------------------------------- group: every partition on its own
    tPhase = Clock::now();
    std::vector<std::vector<Rec>>& precs = W->precs; precs.resize(NPART);       // records of a partition sorted by (hash, name, file order)
    std::vector<std::vector<Unit>> punits(NPART);         // all units of the partition (complete or not)
    std::vector<int64_t> pIncomplete(NPART, 0);
    auto rec_name = [&](const Rec& r) { return r.name(); };
    // Sorted by (hash, file order) on 24-byte integer keys; the NAMES are looked at once per record afterwards, in sorted order and prefetched (they lie in the
    // inflated file, a cache miss each: inside the sort's comparator they cost several misses per record).  Different names with one 64-bit hash -- never seen,
    // possible -- put the run they share into (name, file order) order, as before.
    struct Key { uint64_t hash, order; uint32_t idx; };
    parallel_for(NPART, T, [&](int64_t p, int) {
        std::vector<Rec>& R = precs[(size_t)p];
        size_t n = 0; for(const Arena& a : arenas) n += a.part[(size_t)p].size();
        if(n > 0xFFFFFFFFull) throw Fail("BAM too large: more than 2^32 records in one name partition");
        std::vector<Rec> in; in.reserve(n);
        for(Arena& a : arenas) { std::vector<Rec>& v = a.part[(size_t)p]; in.insert(in.end(), v.begin(), v.end()); std::vector<Rec>().swap(v); }
        std::vector<Key> keys(n);
        for(size_t i = 0; i < n; i++) { keys[i].hash = in[i].hash; keys[i].order = in[i].order; keys[i].idx = (uint32_t)i; }
        std::sort(keys.begin(), keys.end(), [](const Key& a, const Key& b) { return a.hash != b.hash ? a.hash < b.hash : a.order < b.order; });
        auto same_name = [&](const Rec& a, const Rec& b) { return a.nameLen == b.nameLen && memcmp(rec_name(a), rec_name(b), a.nameLen) == 0; };
        std::vector<uint8_t> starts(n + 1, 0);              // starts[i]: record i of the sorted partition is the first of its name
        for(size_t i = 0; i < n;) {
            size_t j = i + 1;
            while(j < n && keys[j].hash == keys[i].hash) j++;
            if(j + 16 < n) __builtin_prefetch(&in[keys[j + 16].idx]);
            if(j + 4 < n) __builtin_prefetch(rec_name(in[keys[j + 4].idx]));
            bool oneName = true;
            for(size_t k = i + 1; k < j && oneName; k++) oneName = same_name(in[keys[k].idx], in[keys[i].idx]);
            starts[i] = 1;
            if(!oneName) {
                std::sort(keys.begin() + (ptrdiff_t)i, keys.begin() + (ptrdiff_t)j, [&](const Key& ka, const Key& kb) {
                    const Rec& a = in[ka.idx]; const Rec& b = in[kb.idx];
                    if(!same_name(a, b)) { const size_t m = a.nameLen < b.nameLen ? a.nameLen : b.nameLen; const int c = memcmp(rec_name(a), rec_name(b), m); return c != 0 ? c < 0 : a.nameLen < b.nameLen; }
                    return ka.order < kb.order;
                });
                for(size_t k = i + 1; k < j; k++) starts[k] = same_name(in[keys[k].idx], in[keys[k - 1].idx]) ? 0 : 1;
            }
            i = j;
        }
        starts[n] = 1;
        R.resize(n);
        for(size_t i = 0; i < n; i++) R[i] = in[keys[i].idx];
        std::vector<Unit>& U = punits[(size_t)p];
        for(size_t i = 0; i < n;) {
            size_t j = i + 1;
            while(!starts[j]) j++;
            bool prim[2] = {false, false};
            for(size_t k = i; k < j; k++) if(R[k].flags & 2) prim[R[k].which] = true;                // takeAlignment, protoSeeds.cpp:23-36
            const bool complete = long_read_mode ? prim[0] : (prim[0] && prim[1]);                  // isComplete / isComplete_unpaired, protoSeeds.cpp:371-380
            if(complete) { Unit u; u.name = rec_name(R[i]); u.nameLen = R[i].nameLen; u.part = (uint32_t)p; u.first = (uint32_t)i; u.count = (uint32_t)(j - i); U.push_back(u); }
            else pIncomplete[(size_t)p]++;
            i = j;
        }
    });
    std::vector<Unit>& units = W->units;
    { size_t n = 0; for(auto& u : punits) n += u.size(); units.reserve(n); for(auto& u : punits) { units.insert(units.end(), u.begin(), u.end()); std::vector<Unit>().swap(u); } }
    for(int p = 0; p < NPART; p++) S->n_incomplete += pIncomplete[(size_t)p];
    S->n_seeds = (int64_t)units.size() + S->n_incomplete;
    S->seconds[3] = since(tPhase);

    // ---------------------------------------------------------------- sort: read-name order (parallel sample sort)
    tPhase = Clock::now();
    if(T > 1 && units.size() > 100000) {
        const int NB = T * 4;
        std::vector<Unit> sample;
        const size_t stepS = std::max<size_t>(1, units.size() / (size_t)(NB * 64));
        for(size_t i = 0; i < units.size(); i += stepS) sample.push_back(units[i]);
        std::sort(sample.begin(), sample.end(), name_less);
        std::vector<Unit> split;
        for(int k = 1; k < NB; k++) split.push_back(sample[sample.size() * (size_t)k / (size_t)NB]);
        const int64_t CHK = 65536; const int64_t nChunks = ((int64_t)units.size() + CHK - 1) / CHK;
        std::vector<uint16_t> bucketOf(units.size());
        std::vector<std::vector<int64_t>> cnt((size_t)nChunks, std::vector<int64_t>((size_t)NB, 0));
        parallel_for(nChunks, T, [&](int64_t c, int) {
            const size_t a = (size_t)(c * CHK), z = std::min(units.size(), a + (size_t)CHK);
            for(size_t i = a; i < z; i++) { const int bk = (int)(std::upper_bound(split.begin(), split.end(), units[i], name_less) - split.begin()); bucketOf[i] = (uint16_t)bk; cnt[(size_t)c][(size_t)bk]++; }
        });
        std::vector<int64_t> bstart((size_t)NB + 1, 0);
        for(int bk = 0; bk < NB; bk++) { int64_t s = 0; for(int64_t c = 0; c < nChunks; c++) s += cnt[(size_t)c][(size_t)bk]; bstart[(size_t)bk + 1] = bstart[(size_t)bk] + s; }
        // per chunk and bucket: where its units go
        { std::vector<int64_t> run(bstart.begin(), bstart.end() - 1); for(int64_t c = 0; c < nChunks; c++) for(int bk = 0; bk < NB; bk++) { const int64_t k = cnt[(size_t)c][(size_t)bk]; cnt[(size_t)c][(size_t)bk] = run[(size_t)bk]; run[(size_t)bk] += k; } }
        std::vector<Unit> sorted(units.size());
        parallel_for(nChunks, T, [&](int64_t c, int) {
            const size_t a = (size_t)(c * CHK), z = std::min(units.size(), a + (size_t)CHK);
            std::vector<int64_t>& at = cnt[(size_t)c];
            for(size_t i = a; i < z; i++) sorted[(size_t)at[bucketOf[i]]++] = units[i];
        });
        parallel_for(NB, T, [&](int64_t bk, int) { std::sort(sorted.begin() + bstart[(size_t)bk], sorted.begin() + bstart[(size_t)bk + 1], name_less); });
        units.swap(sorted);
    } else std::sort(units.begin(), units.end(), name_less);
    S->seconds[4] = since(tPhase);

    // ---------------------------------------------------------------- layout
    tPhase = Clock::now();
    const int nm = long_read_mode ? 1 : 2;
    const size_t nU = units.size(), nR = nU * (size_t)nm;
    S->n_units = (int64_t)nU;
    // (offset arrays of 8 bytes per read: not cleared -- every entry but the first is written by the pass below, read_primary by the fill)
    S->read_off.alloc(nR + 1); S->chain_off.alloc(nR + 1); S->name_off.alloc(nU + 1); S->read_off[0] = 0; S->chain_off[0] = 0; S->name_off[0] = 0;
    S->read_primary.alloc(nR + 1);
    // sizes per read: bases, chains, cigar operations; the primary of a mate is only known after its sort, so the alignments of every mate are
    // ordered here once (kept as record indices) and reused by the fill pass
    Buf<int64_t>& cigCount = W->cigCount; cigCount.alloc(nR + 1); cigCount[0] = 0;
    std::vector<std::vector<uint32_t>> order((size_t)T);
    const int64_t UCH = std::max<int64_t>(16, std::min<int64_t>(8192, (int64_t)nU / ((int64_t)T * 8) + 1)); const int64_t nUChunks = ((int64_t)nU + UCH - 1) / UCH;
    parallel_for(nUChunks, T, [&](int64_t c, int t) {
        std::vector<uint32_t>& idx = order[(size_t)t];
        const size_t a = (size_t)(c * UCH), z = std::min(nU, a + (size_t)UCH);
        for(size_t ui = a; ui < z; ui++) {
            // (the units are walked in name order, their records sit wherever the name hash put them: the descriptors of the unit eight ahead are requested now)
            if(ui + 8 < z) { const Unit& u8 = units[ui + 8]; const Rec* r8 = precs[u8.part].data() + u8.first; for(uint32_t k = 0; k < u8.count; k += 1) __builtin_prefetch(r8 + k); }
            const Unit& u = units[ui]; const std::vector<Rec>& R = precs[u.part];
            S->name_off[ui + 1] = (int64_t)u.nameLen + 1;
            for(int m = 0; m < nm; m++) {
                const size_t prim = sorted_mate(precs, u, m, idx);
                const size_t r = ui * (size_t)nm + (size_t)m;
                S->read_off[r + 1] = R[idx[prim]].l_seq; S->chain_off[r + 1] = (int64_t)idx.size();
                int64_t cg = 0; for(uint32_t k : idx) cg += R[k].n_cigar;
                cigCount[r + 1] = cg;
            }
        }
    });
    const bool dbgL = getenv("HLALA_BAM_DEBUG") != nullptr; const double tL1 = since(tPhase);
    for(size_t r = 0; r < nR; r++) { S->read_off[r + 1] += S->read_off[r]; S->chain_off[r + 1] += S->chain_off[r]; cigCount[r + 1] += cigCount[r]; }
    for(size_t ui = 0; ui < nU; ui++) S->name_off[ui + 1] += S->name_off[ui];
    const size_t nBases = (size_t)S->read_off[nR], nChains = (size_t)S->chain_off[nR], nCig = (size_t)cigCount[nR];
    if(nChains > 0x7FFFFFFFull) throw Fail("more than 2^31 - 1 alignments in one sample: chain numbers are 32-bit");
    if(S->packed) S->read_bases_packed.alloc((nBases + nR + 1) / 2 + 2); else S->read_bases.alloc(nBases);
    S->read_quals.alloc(nBases);
    S->chain_contig.alloc(nChains); S->chain_pos.alloc(nChains); S->chain_offset.alloc(nChains); S->chain_as.alloc(nChains); S->chain_reverse.alloc(nChains);
    S->cigar_off.alloc(nChains + 1); S->cigar_off[0] = 0; S->cigar.alloc(nCig); S->name_chars.alloc((size_t)S->name_off[nU]);
    const double tL2 = since(tPhase);
    // ---- the fill.  Names, bases, qualities, alignments and CIGARs of a unit are written when a window that holds it is first handed out
    // (hlala_seed_batch_window / _desc / _name -> ensure_filled): a caller that walks the sample batch by batch has the GPU align one batch while the host
    // fills the next, instead of waiting for the whole sample here (0.6 s of 2.9 for 8.4 M pairs).  HLALA_BAM_EAGER=1 fills everything now.
    W->nm = nm; W->T = T; W->UCH = UCH; W->nUChunks = nUChunks; W->remaining = nUChunks; W->done.assign((size_t)nUChunks, 0);
    S->seconds[5] = since(tPhase);
    if(dbgL) fprintf(stderr, "bam-debug: layout: sizes %.3f, prefix sums + allocation %.3f s; the fill runs per window\n", tL1, tL2 - tL1);
    S->work = W.release();
    if(nUChunks == 0) { delete S->work; S->work = nullptr; }
    else if(const char* e = getenv("HLALA_BAM_EAGER")) { if(atoi(e) != 0) ensure_filled(S.get(), 0, (int64_t)nU); }
    *out = S.release();
    return HLALA_OK;
} catch(const std::exception& e_) { g_bam_error = dynamic_cast<const Fail*>(&e_) ? std::string(e_.what()) : std::string("hlala_bam_extract_seeds: ") + e_.what(); return HLALA_E_ARG; }

extern "C" int hlala_seed_batch_window(const hlala_seed_batch* S, int64_t first_unit, int32_t n_units, hlala_batch_in* in)
{
    if(!S || !in || first_unit < 0 || n_units < 0 || first_unit + n_units > S->n_units) { g_bam_error = "hlala_seed_batch_window: units outside the sample"; return HLALA_E_ARG; }
    const int per = S->unpaired ? 1 : 2;
    const size_t r0 = (size_t)first_unit * (size_t)per, nr = (size_t)n_units * (size_t)per;
    try { ensure_filled(S, first_unit, first_unit + n_units); } catch(const std::exception& e_) { g_bam_error = std::string("hlala_seed_batch_window: ") + e_.what(); return HLALA_E_ARG; }
    if(S->pin_lazy && hlala_host::g_seed_batch_pin_upto) hlala_host::g_seed_batch_pin_upto(const_cast<hlala_seed_batch*>(S), first_unit + n_units);      // (filled first: the pages exist when they are locked)
    const int64_t nb = S->read_off[r0 + nr] - S->read_off[r0], nc = S->chain_off[r0 + nr] - S->chain_off[r0];
    const int64_t ng = S->cigar_off[(size_t)S->chain_off[r0 + nr]] - S->cigar_off[(size_t)S->chain_off[r0]];
    if(nb > 0x7FFFFFFFll || nc > 0x7FFFFFFFll || ng > 0x7FFFFFFFll) {
        g_bam_error = "hlala_seed_batch_window: " + std::to_string(n_units) + " units hold " + std::to_string(nb) + " bases / " + std::to_string(nc) + " alignments: more than one batch holds (2^31 - 1)";
        return HLALA_E_CAPACITY;
    }
    in->n_pairs = n_units; in->read_off = S->read_off.data() + r0; in->read_bases = S->read_bases.data(); in->read_quals = S->read_quals.data();
    in->read_bases_packed = S->packed ? S->read_bases_packed.data() : nullptr; in->first_read = (int64_t)r0;
    in->chain_off = S->chain_off.data() + r0; in->read_primary = S->read_primary.data() + r0; in->n_chains = (int32_t)nc;
    in->chain_contig = S->chain_contig.data(); in->chain_pos = S->chain_pos.data(); in->chain_offset = S->chain_offset.data(); in->chain_as = S->chain_as.data();
    in->chain_reverse = S->chain_reverse.data(); in->cigar_off = S->cigar_off.data(); in->cigar = S->cigar.data();
    return HLALA_OK;
}

extern "C" int hlala_seed_batch_desc(const hlala_seed_batch* S, hlala_batch_in* in, int64_t* counts /* [3] examined records, seeds, incomplete seeds; or NULL */)
{
    if(!S || !in) return HLALA_E_ARG;
    if(S->n_units > 0x7FFFFFFFll) { g_bam_error = "hlala_seed_batch_desc: more than 2^31 - 1 units"; return HLALA_E_CAPACITY; }
    try { ensure_filled(S, 0, S->n_units); } catch(const std::exception& e_) { g_bam_error = std::string("hlala_seed_batch_desc: ") + e_.what(); return HLALA_E_ARG; }
    // the whole sample: n_chains saturates when the sample holds more than one batch can (hlala_batch_create refuses such a descriptor;
    // hlala_seed_batch_window cuts batches)
    in->n_pairs = (int32_t)S->n_units; in->read_off = S->read_off.data(); in->read_bases = S->read_bases.data(); in->read_quals = S->read_quals.data();
    in->read_bases_packed = S->packed ? S->read_bases_packed.data() : nullptr; in->first_read = 0;
    in->chain_off = S->chain_off.data(); in->read_primary = S->read_primary.data(); in->n_chains = (int32_t)std::min<size_t>(S->chain_contig.size(), 0x7FFFFFFF);
    in->chain_contig = S->chain_contig.data(); in->chain_pos = S->chain_pos.data(); in->chain_offset = S->chain_offset.data(); in->chain_as = S->chain_as.data();
    in->chain_reverse = S->chain_reverse.data(); in->cigar_off = S->cigar_off.data(); in->cigar = S->cigar.data();
    if(counts) { counts[0] = S->examined; counts[1] = S->n_seeds; counts[2] = S->n_incomplete; }
    return HLALA_OK;
}

extern "C" int64_t hlala_seed_batch_units(const hlala_seed_batch* S) { return S ? S->n_units : 0; }
extern "C" int hlala_seed_batch_counts(const hlala_seed_batch* S, int64_t* counts)
{
    if(!S || !counts) return HLALA_E_ARG;
    counts[0] = S->examined; counts[1] = S->n_seeds; counts[2] = S->n_incomplete;
    return HLALA_OK;
}
extern "C" const char* hlala_seed_batch_name(const hlala_seed_batch* S, int64_t unit)
{
    if(!S || unit < 0 || unit >= S->n_units) return nullptr;
    try { ensure_filled(S, unit, unit + 1); } catch(...) { return nullptr; }
    return S->name_chars.data() + S->name_off[(size_t)unit];
}
extern "C" int hlala_seed_batch_timing(const hlala_seed_batch* S, double* seconds6, int32_t* n_threads)
{
    if(!S) return HLALA_E_ARG;
    if(seconds6) for(int i = 0; i < 6; i++) seconds6[i] = S->seconds[i];
    if(n_threads) *n_threads = S->threads;
    return HLALA_OK;
}
extern "C" void hlala_seed_batch_free(hlala_seed_batch* S)
{
    if(S && S->pinned && hlala_host::g_seed_batch_unpin) hlala_host::g_seed_batch_unpin(S);
    delete S;
}
