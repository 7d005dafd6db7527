// bamwriter.cpp -- a fast BAM writer for synthetic samples (test and bench infrastructure, not part of the product): records given as
// flat arrays are encoded following the SAM/BAM specification v1 (little-endian records, 4-bit bases, BGZF blocks = gzip members with a
// BC extra field) and deflated on all host threads.  The product's reader (hla-la_amd/csrc/host_bam.cpp) is tested against files written
// here and against files written by the independent pure-Python writer of tests/test_bam.py.
#include <algorithm>
#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include <zlib.h>

namespace {

struct Writer {
    FILE* f = nullptr; int threads = 1, level = 1; std::string err;
    std::vector<uint8_t> pending;        // uncompressed bytes not yet written (less than one block is kept between appends)
    long long bytes_out = 0, records = 0;
};

thread_local std::string g_err;

constexpr size_t BLOCK = 65280;          // uncompressed payload per BGZF block (htslib's choice)

void put32(std::vector<uint8_t>& v, uint32_t x) { v.push_back(x & 255); v.push_back((x >> 8) & 255); v.push_back((x >> 16) & 255); v.push_back((x >> 24) & 255); }
void put16(std::vector<uint8_t>& v, uint32_t x) { v.push_back(x & 255); v.push_back((x >> 8) & 255); }

// one BGZF block from [p, p + n)
bool bgzf_block(const uint8_t* p, size_t n, int level, std::vector<uint8_t>& out)
{
    out.clear();
    std::vector<uint8_t> comp(compressBound((uLong)n) + 64);
    z_stream zs; memset(&zs, 0, sizeof(zs));
    if(deflateInit2(&zs, level, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) return false;
    zs.next_in = (Bytef*)p; zs.avail_in = (uInt)n; zs.next_out = comp.data(); zs.avail_out = (uInt)comp.size();
    const int rc = deflate(&zs, Z_FINISH); const size_t cn = zs.total_out; deflateEnd(&zs);
    if(rc != Z_STREAM_END) return false;
    const size_t bsize = cn + 25;                 // header 18 + payload + crc 4 + isize 4, minus 1
    if(bsize > 65535) return false;
    const uint8_t hdr[12] = {31, 139, 8, 4, 0, 0, 0, 0, 0, 255, 6, 0};
    out.insert(out.end(), hdr, hdr + 12); out.push_back('B'); out.push_back('C'); put16(out, 2); put16(out, (uint32_t)bsize);
    out.insert(out.end(), comp.begin(), comp.begin() + (long)cn);
    put32(out, (uint32_t)crc32(crc32(0L, Z_NULL, 0), p, (uInt)n)); put32(out, (uint32_t)n);
    return true;
}

// compress and write all complete blocks of `pending` (everything if `all`)
bool flush(Writer* w, bool all)
{
    const size_t n = w->pending.size();
    const size_t nBlocks = all ? (n + BLOCK - 1) / BLOCK : n / BLOCK;
    if(nBlocks == 0) return true;
    std::vector<std::vector<uint8_t>> outs(nBlocks);
    std::atomic<size_t> next(0); std::atomic<bool> ok(true);
    auto work = [&]() { for(;;) { size_t b = next.fetch_add(1); if(b >= nBlocks) break; const size_t a = b * BLOCK, z = std::min(n, a + BLOCK); if(!bgzf_block(w->pending.data() + a, z - a, w->level, outs[b])) ok = false; } };
    std::vector<std::thread> th; const int T = (int)std::min<size_t>((size_t)w->threads, nBlocks);
    for(int t = 1; t < T; t++) th.emplace_back(work);
    work();
    for(auto& x : th) x.join();
    if(!ok) { w->err = "deflate failed"; return false; }
    for(auto& o : outs) { if(fwrite(o.data(), 1, o.size(), w->f) != o.size()) { w->err = "write failed"; return false; } w->bytes_out += (long long)o.size(); }
    const size_t used = std::min(n, nBlocks * BLOCK);
    w->pending.erase(w->pending.begin(), w->pending.begin() + (long)used);
    return true;
}

}  // namespace

extern "C" {

const char* bw_last_error() { return g_err.c_str(); }

void* bw_open(const char* path, int n_refs, const char* const* ref_names, const int32_t* ref_lengths, int threads, int level)
{
    Writer* w = new Writer();
    w->f = fopen(path, "wb");
    if(!w->f) { g_err = std::string("cannot open ") + path; delete w; return nullptr; }
    w->threads = threads > 0 ? threads : (int)std::max(1u, std::thread::hardware_concurrency()); w->level = level;
    std::vector<uint8_t>& v = w->pending;
    v.insert(v.end(), {'B', 'A', 'M', 1}); put32(v, 0); put32(v, (uint32_t)n_refs);
    for(int i = 0; i < n_refs; i++) { const size_t l = strlen(ref_names[i]) + 1; put32(v, (uint32_t)l); v.insert(v.end(), ref_names[i], ref_names[i] + l); put32(v, (uint32_t)ref_lengths[i]); }
    return w;
}

// n records: name r = name_chars[name_off[r] .. name_off[r+1]) (no terminator), flag, reference id, 0-based position, CIGAR operations
// cigar[cigar_off[r] .. cigar_off[r+1]) in BAM encoding, bases (ASCII over =ACMGRSVTWYHKDBN) and qualities (Phred + 33) at
// [seq_off[r], seq_off[r+1]) (empty: SEQ '*', as bwa writes secondary alignments), AS tag value.  Records are written in the given order.
int bw_append(void* h, int64_t n, const char* name_chars, const int64_t* name_off, const uint16_t* flag, const int32_t* ref, const int32_t* pos,
              const int64_t* cigar_off, const uint32_t* cigar, const int64_t* seq_off, const uint8_t* bases, const uint8_t* quals, const int32_t* as)
{
    Writer* w = (Writer*)h;
    if(!w || n < 0) return -1;
    static uint8_t code[256]; static bool init = false;
    if(!init) { memset(code, 15, sizeof(code)); const char* S = "=ACMGRSVTWYHKDBN"; for(int i = 0; i < 16; i++) code[(unsigned char)S[i]] = (uint8_t)i; init = true; }
    // sizes, then parallel encoding into one buffer
    std::vector<int64_t> off((size_t)n + 1, 0);
    for(int64_t r = 0; r < n; r++) {
        const int64_t ln = name_off[r + 1] - name_off[r] + 1, nc = cigar_off[r + 1] - cigar_off[r], ls = seq_off[r + 1] - seq_off[r];
        if(ln > 255 || nc > 65535) { g_err = "record name or CIGAR too long"; return -1; }
        off[(size_t)r + 1] = off[(size_t)r] + 4 + 32 + ln + 4 * nc + (ls + 1) / 2 + ls + 7;       // + AS:i tag (3 + 4 bytes)
    }
    const size_t base = w->pending.size();
    w->pending.resize(base + (size_t)off[(size_t)n]);
    uint8_t* out = w->pending.data() + base;
    std::atomic<int64_t> next(0); const int64_t CH = 16384;
    auto work = [&]() {
        for(;;) {
            const int64_t r0 = next.fetch_add(CH); if(r0 >= n) break;
            const int64_t r1 = std::min(n, r0 + CH);
            for(int64_t r = r0; r < r1; r++) {
                uint8_t* p = out + off[(size_t)r];
                const int64_t ln = name_off[r + 1] - name_off[r] + 1, nc = cigar_off[r + 1] - cigar_off[r], ls = seq_off[r + 1] - seq_off[r];
                auto w32 = [&](uint32_t x) { p[0] = x & 255; p[1] = (x >> 8) & 255; p[2] = (x >> 16) & 255; p[3] = (x >> 24) & 255; p += 4; };
                auto w16 = [&](uint32_t x) { p[0] = x & 255; p[1] = (x >> 8) & 255; p += 2; };
                w32((uint32_t)(off[(size_t)r + 1] - off[(size_t)r] - 4));
                w32((uint32_t)ref[r]); w32((uint32_t)pos[r]); *p++ = (uint8_t)ln; *p++ = 60; w16(4680); w16((uint32_t)nc); w16(flag[r]); w32((uint32_t)ls);
                w32(0xFFFFFFFFu); w32(0xFFFFFFFFu); w32(0);
                memcpy(p, name_chars + name_off[r], (size_t)ln - 1); p += ln - 1; *p++ = 0;
                for(int64_t k = 0; k < nc; k++) w32(cigar[cigar_off[r] + k]);
                const uint8_t* b = bases + seq_off[r]; const uint8_t* q = quals + seq_off[r];
                for(int64_t i = 0; i + 1 < ls; i += 2) *p++ = (uint8_t)((code[b[i]] << 4) | code[b[i + 1]]);
                if(ls & 1) *p++ = (uint8_t)(code[b[ls - 1]] << 4);
                for(int64_t i = 0; i < ls; i++) *p++ = (uint8_t)(q[i] - 33);
                *p++ = 'A'; *p++ = 'S'; *p++ = 'i'; w32((uint32_t)as[r]);
            }
        }
    };
    std::vector<std::thread> th; for(int t = 1; t < w->threads; t++) th.emplace_back(work);
    work();
    for(auto& x : th) x.join();
    w->records += n;
    if(!flush(w, false)) { g_err = w->err; return -1; }
    return 0;
}

// bytes written so far (after close: the file size)
long long bw_close(void* h)
{
    Writer* w = (Writer*)h;
    if(!w) return -1;
    long long total = -1;
    std::vector<uint8_t> eofb;
    if(flush(w, true) && bgzf_block(nullptr, 0, w->level, eofb) && fwrite(eofb.data(), 1, eofb.size(), w->f) == eofb.size()) total = w->bytes_out + (long long)eofb.size();
    else g_err = w->err.empty() ? "write failed" : w->err;
    if(fclose(w->f) != 0) total = -1;
    delete w;
    return total;
}

}  // extern "C"
