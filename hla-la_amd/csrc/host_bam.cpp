// host_bam.cpp -- BAM reader + seed extraction: the records of a bwa-mem BAM that fall into the PRG intervals become the
// pairs / chains of an hlala_batch_in.
//
// Reference: processBAM::extractSeeds2 (mapper/processBAM.cpp:703-864), reads::protoSeeds::takeAlignment / isComplete
// (mapper/reads/protoSeeds.cpp:23-36, 371-380), sortChainsInSeeds (:1945-1967), getAlignmentScore (:4314-4334: the AS tag), the order of
// completeProtoSeeds (std::map over read names, :2024-2039).  BamTools (un-vendored, "tested with 2.5.1", makefile:3-12) is replaced by
// a direct reading of the BAM format (SAM/BAM specification v1: BGZF blocks = gzip members with a BC extra field, little-endian records).
//   * a record is used if it is mapped, (long-read mode: primary,) its reference carries intervals, it has CIGAR operations, and both its
//     start and its end (Position + reference-consuming length - 1 = GetEndPosition(false, true)) lie inside an interval (:763-768);
//   * positions are re-based to the interval start (the reference passes the interval start as reference2level_offset_0based and indexes
//     the translation with position - offset; here chain_pos = position - start and chain_offset = 0, the contig is the interval);
//   * std::sort + std::reverse on the alignment scores is the reference's own call (:1952-1961): equal scores keep the library's order.
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <unordered_map>
#include <vector>

#include <zlib.h>

#include "../../include/hlala_gpu.h"

namespace {

thread_local std::string g_bam_error;

// ---- BGZF: concatenated gzip members, each at most 64 KiB of payload
struct Bgzf {
    FILE* f = nullptr; std::vector<uint8_t> in, out; size_t pos = 0; bool eof = false;
    bool fill()
    {
        uint8_t h[18];
        size_t n = fread(h, 1, 18, f);
        if(n == 0) { eof = true; return false; }
        if(n != 18 || h[0] != 31 || h[1] != 139 || h[2] != 8 || !(h[3] & 4)) { g_bam_error = "not a BGZF block"; return false; }
        const unsigned xlen = h[10] | (h[11] << 8);
        // the BC subfield holds the block size - 1; it is the first subfield in every BGZF writer, but walk the extra field anyway
        std::vector<uint8_t> extra(xlen); memcpy(extra.data(), h + 12, std::min<size_t>(6, xlen));
        if(xlen > 6 && fread(extra.data() + 6, 1, xlen - 6, f) != xlen - 6) { g_bam_error = "truncated BGZF header"; return false; }
        int bsize = -1;
        for(size_t p = 0; p + 4 <= xlen;) { unsigned sl = extra[p + 2] | (extra[p + 3] << 8); if(extra[p] == 66 && extra[p + 1] == 67 && sl == 2 && p + 6 <= xlen) bsize = extra[p + 4] | (extra[p + 5] << 8); p += 4 + sl; }
        if(bsize < 0) { g_bam_error = "BGZF block without BC field"; return false; }
        if((size_t)bsize + 1 < (size_t)12 + xlen + 8) { g_bam_error = "BGZF block size smaller than its own header"; return false; }
        const size_t cdata = (size_t)bsize + 1 - 12 - xlen - 8;
        in.resize(cdata + 8);
        if(fread(in.data(), 1, cdata + 8, f) != cdata + 8) { g_bam_error = "truncated BGZF block"; return false; }
        const uint32_t isize = in[cdata + 4] | (in[cdata + 5] << 8) | (in[cdata + 6] << 16) | ((uint32_t)in[cdata + 7] << 24);
        out.resize(isize); pos = 0;
        if(isize == 0) return true;                                   // the empty end-of-file block
        z_stream zs; memset(&zs, 0, sizeof(zs));
        if(inflateInit2(&zs, -15) != Z_OK) { g_bam_error = "inflateInit2 failed"; return false; }
        zs.next_in = in.data(); zs.avail_in = (uInt)cdata; zs.next_out = out.data(); zs.avail_out = isize;
        int rc = inflate(&zs, Z_FINISH); inflateEnd(&zs);
        if(rc != Z_STREAM_END || zs.total_out != isize) { g_bam_error = "BGZF inflate failed"; return false; }
        return true;
    }
    bool read(void* dst, size_t n)
    {
        uint8_t* d = (uint8_t*)dst;
        while(n) {
            if(pos == out.size()) { if(!fill()) return false; continue; }
            size_t k = std::min(n, out.size() - pos);
            memcpy(d, out.data() + pos, k); d += k; pos += k; n -= k;
        }
        return true;
    }
    bool at_end() { while(pos == out.size()) { if(!fill()) return true; } return false; }
};

struct Aln { int contig, pos, as; bool reverse, primary; std::vector<uint32_t> cigar; std::string bases, quals; };
struct Proto { std::vector<Aln> r[2]; bool havePrimary[2] = {false, false}; };

uint32_t rd32(const uint8_t* p) { return p[0] | (p[1] << 8) | (p[2] << 16) | ((uint32_t)p[3] << 24); }

}  // namespace

struct hlala_seed_batch {
    std::vector<int32_t> read_off, chain_off, read_primary, chain_contig, chain_pos, chain_offset, chain_as, cigar_off;
    std::vector<uint8_t> read_bases, read_quals, chain_reverse; std::vector<uint32_t> cigar;
    std::vector<std::string> names;
    int32_t n_units = 0, unpaired = 0; int64_t examined = 0, n_seeds = 0, n_incomplete = 0;
};

extern "C" const char* hlala_bam_last_error() { return g_bam_error.c_str(); }

extern "C" int hlala_bam_extract_seeds(const char* path, int32_t n_intervals, const hlala_bam_interval* iv, int32_t long_read_mode, hlala_seed_batch** out)
try {
    if(!path || !out || n_intervals < 0 || (n_intervals > 0 && !iv)) return HLALA_E_ARG;
    *out = nullptr; g_bam_error.clear();
    Bgzf z; z.f = fopen(path, "rb");
    if(!z.f) { g_bam_error = std::string("Cannot open BAM file: ") + path; return HLALA_E_ARG; }
    auto fail = [&](const std::string& m) { if(g_bam_error.empty()) g_bam_error = m; fclose(z.f); return HLALA_E_ARG; };
    char magic[4]; int32_t l_text = 0, n_ref = 0;
    if(!z.read(magic, 4) || memcmp(magic, "BAM\1", 4) != 0) return fail("not a BAM file");
    if(!z.read(&l_text, 4) || l_text < 0 || l_text > (1 << 30)) return fail("truncated BAM header");
    { std::vector<char> text((size_t)l_text); if(l_text && !z.read(text.data(), (size_t)l_text)) return fail("truncated BAM header text"); }
    if(!z.read(&n_ref, 4) || n_ref < 0) return fail("truncated BAM header");
    std::vector<std::string> refName((size_t)n_ref);
    for(int i = 0; i < n_ref; i++) {
        int32_t l_name = 0, l_ref = 0;
        if(!z.read(&l_name, 4) || l_name < 1 || l_name > (1 << 20)) return fail("truncated BAM reference list");
        std::vector<char> nm((size_t)l_name); if(!z.read(nm.data(), (size_t)l_name) || !z.read(&l_ref, 4)) return fail("truncated BAM reference list");
        refName[i] = std::string(nm.data());
    }
    std::unordered_map<std::string, std::vector<int>> intervalsOfRef;                     // interestingIntervals, processBAM.cpp:1226-1400
    for(int i = 0; i < n_intervals; i++) { if(!iv[i].ref_name || iv[i].stop_0based < iv[i].start_0based) return fail("bad interval"); intervalsOfRef[iv[i].ref_name].push_back(i); }
    std::map<std::string, Proto> seeds;                                                    // std::map: read-name order, :712
    int64_t examined = 0;
    std::vector<uint8_t> rec;
    static const char SEQ16[] = "=ACMGRSVTWYHKDBN";
    while(!z.at_end()) {
        int32_t block_size = 0;
        if(!z.read(&block_size, 4) || block_size < 32 || block_size > (1 << 28)) return fail("truncated BAM record");
        rec.resize((size_t)block_size);
        if(!z.read(rec.data(), (size_t)block_size)) return fail("truncated BAM record");
        const int32_t refID = (int32_t)rd32(&rec[0]), position = (int32_t)rd32(&rec[4]);
        const unsigned l_read_name = rec[8]; const unsigned n_cigar = rec[12] | (rec[13] << 8); const unsigned flag = rec[14] | (rec[15] << 8);
        const int32_t l_seq = (int32_t)rd32(&rec[16]);
        const size_t oName = 32, oCigar = oName + l_read_name, oSeq = oCigar + 4 * (size_t)n_cigar, oQual = oSeq + ((size_t)l_seq + 1) / 2, oTags = oQual + (size_t)l_seq;
        if(oTags > rec.size()) return fail("corrupt BAM record");
        if(flag & 4) continue;                                                             // ! IsMapped(), :727
        if(long_read_mode && (flag & 256)) continue;                                       // ! IsPrimaryAlignment(), :732-738 (BamTools 2.5.1: !(AlignmentFlag & 0x100))
        if(refID < 0 || refID >= n_ref) continue;
        auto ivs = intervalsOfRef.find(refName[(size_t)refID]);
        if(ivs == intervalsOfRef.end()) continue;                                          // :744
        for(int ii : ivs->second) {
            examined++;                                                                    // :757 (per interval, as in the reference)
            if(n_cigar == 0) continue;                                                     // :759-763
            int refLen = 0;
            for(unsigned k = 0; k < n_cigar; k++) { uint32_t c = rd32(&rec[oCigar + 4 * k]); unsigned op = c & 15u; if(op == 0 || op == 2 || op == 3 || op == 7 || op == 8) refLen += (int)(c >> 4); }
            const int start = position, stop = position + refLen - 1;                      // GetEndPosition(false, true), :766
            if(!((start >= iv[ii].start_0based && start <= iv[ii].stop_0based) && (stop >= iv[ii].start_0based && stop <= iv[ii].stop_0based))) continue;
            if(!long_read_mode && !(flag & 1)) return fail("unpaired record in a paired-end BAM (assert(currentAlignment.IsPaired()), processBAM.cpp:783)");
            Aln a; a.contig = iv[ii].contig; a.pos = position - iv[ii].start_0based; a.reverse = (flag & 16) != 0; a.primary = !(flag & 256);     // BamTools IsPrimaryAlignment: !(flag & 0x100)
            a.cigar.resize(n_cigar); for(unsigned k = 0; k < n_cigar; k++) a.cigar[k] = rd32(&rec[oCigar + 4 * k]);
            a.bases.resize((size_t)l_seq); a.quals.resize((size_t)l_seq);
            for(int32_t i = 0; i < l_seq; i++) { unsigned b = rec[oSeq + (size_t)i / 2]; a.bases[(size_t)i] = SEQ16[(i & 1) ? (b & 15) : (b >> 4)]; a.quals[(size_t)i] = (char)(rec[oQual + (size_t)i] + 33); }   // BuildCharData: Phred + 33
            // the AS tag (getAlignmentScore, :4314-4334): any integer type
            bool haveAS = false; a.as = 0;
            for(size_t p = oTags; p + 3 <= rec.size();) {
                const char t0 = (char)rec[p], t1 = (char)rec[p + 1], ty = (char)rec[p + 2]; p += 3;
                size_t sz = 0; long long v = 0; bool isInt = true;
                switch(ty) {
                    case 'c': sz = 1; if(p + 1 <= rec.size()) v = (int8_t)rec[p]; break;
                    case 'C': sz = 1; if(p + 1 <= rec.size()) v = rec[p]; break;
                    case 's': sz = 2; if(p + 2 <= rec.size()) v = (int16_t)(rec[p] | (rec[p + 1] << 8)); break;
                    case 'S': sz = 2; if(p + 2 <= rec.size()) v = (uint16_t)(rec[p] | (rec[p + 1] << 8)); break;
                    case 'i': sz = 4; if(p + 4 <= rec.size()) v = (int32_t)rd32(&rec[p]); break;
                    case 'I': sz = 4; if(p + 4 <= rec.size()) v = rd32(&rec[p]); break;
                    case 'A': sz = 1; isInt = false; break;
                    case 'f': sz = 4; isInt = false; break;
                    case 'Z': case 'H': { isInt = false; size_t q = p; while(q < rec.size() && rec[q]) q++; sz = q - p + 1; break; }
                    case 'B': { isInt = false; if(p + 5 > rec.size()) return fail("corrupt BAM tag"); char et = (char)rec[p]; uint32_t cnt = rd32(&rec[p + 1]);
                                size_t es = (et == 'c' || et == 'C') ? 1 : (et == 's' || et == 'S') ? 2 : 4; sz = 5 + es * (size_t)cnt; break; }
                    default: return fail("unknown BAM tag type");
                }
                if(p + sz > rec.size()) return fail("corrupt BAM tag");
                if(t0 == 'A' && t1 == 'S' && isInt) { a.as = (int)v; haveAS = true; }
                p += sz;
            }
            if(!haveAS) return fail("Can't get AS tag!");                                   // assert(1 == 0), :4330-4332
            const std::string name((const char*)&rec[oName]);
            const int which = long_read_mode ? 0 : ((flag & 64) ? 0 : 1);                  // IsFirstMate() ? 1 : 2; long reads: 1 (:814-818)
            Proto& P = seeds[name];
            P.havePrimary[which] = P.havePrimary[which] || a.primary;                      // takeAlignment, protoSeeds.cpp:23-36
            P.r[which].push_back(std::move(a));
        }
    }
    fclose(z.f);
    hlala_seed_batch* S = new hlala_seed_batch();
    S->examined = examined; S->n_seeds = (int64_t)seeds.size(); S->unpaired = long_read_mode ? 1 : 0;
    S->read_off.push_back(0); S->chain_off.push_back(0); S->cigar_off.push_back(0);
    const int nm = long_read_mode ? 1 : 2;
    for(auto& kv : seeds) {
        Proto& P = kv.second;
        const bool complete = long_read_mode ? P.havePrimary[0] : (P.havePrimary[0] && P.havePrimary[1]);      // isComplete / isComplete_unpaired, protoSeeds.cpp:371-380
        if(!complete) { S->n_incomplete++; continue; }
        S->names.push_back(kv.first);
        for(int m = 0; m < nm; m++) {
            std::vector<Aln>& al = P.r[m];
            std::sort(al.begin(), al.end(), [](const Aln& a, const Aln& b) { return a.as < b.as; });            // sortChainsInSeeds, :1952-1961
            std::reverse(al.begin(), al.end());
            size_t prim = al.size();
            for(size_t i = 0; i < al.size(); i++) if(al[i].primary) { prim = i; break; }                         // read*_getPrimaryAlignmentI: the first primary
            const Aln& pa = al[prim];
            S->read_bases.insert(S->read_bases.end(), pa.bases.begin(), pa.bases.end());                       // QueryBases / Qualities of the primary, alignment orientation (:3142-3145)
            S->read_quals.insert(S->read_quals.end(), pa.quals.begin(), pa.quals.end());
            S->read_off.push_back((int32_t)S->read_bases.size());
            S->read_primary.push_back((int32_t)(S->chain_contig.size() + prim));
            for(const Aln& a : al) {
                S->chain_contig.push_back(a.contig); S->chain_pos.push_back(a.pos); S->chain_offset.push_back(0); S->chain_as.push_back(a.as); S->chain_reverse.push_back(a.reverse ? 1 : 0);
                S->cigar.insert(S->cigar.end(), a.cigar.begin(), a.cigar.end()); S->cigar_off.push_back((int32_t)S->cigar.size());
            }
            S->chain_off.push_back((int32_t)S->chain_contig.size());
        }
        S->n_units++;
    }
    *out = S;
    return HLALA_OK;
} catch(const std::exception& e_) { g_bam_error = std::string("hlala_bam_extract_seeds: ") + e_.what(); return HLALA_E_ARG; }

extern "C" int hlala_seed_batch_desc(const hlala_seed_batch* S, hlala_batch_in* in, int64_t* counts /* [3] examined records, seeds, incomplete seeds; or NULL */)
try {
    if(!S || !in) return HLALA_E_ARG;
    in->n_pairs = S->n_units; in->read_off = S->read_off.data(); in->read_bases = S->read_bases.data(); in->read_quals = S->read_quals.data();
    in->chain_off = S->chain_off.data(); in->read_primary = S->read_primary.data(); in->n_chains = (int32_t)S->chain_contig.size();
    in->chain_contig = S->chain_contig.data(); in->chain_pos = S->chain_pos.data(); in->chain_offset = S->chain_offset.data(); in->chain_as = S->chain_as.data();
    in->chain_reverse = S->chain_reverse.data(); in->cigar_off = S->cigar_off.data(); in->cigar = S->cigar.data();
    if(counts) { counts[0] = S->examined; counts[1] = S->n_seeds; counts[2] = S->n_incomplete; }
    return HLALA_OK;
} catch(const std::exception& e_) { g_bam_error = std::string("hlala_seed_batch_desc: ") + e_.what(); return HLALA_E_ARG; }

extern "C" const char* hlala_seed_batch_name(const hlala_seed_batch* S, int32_t unit) { return (S && unit >= 0 && unit < S->n_units) ? S->names[(size_t)unit].c_str() : nullptr; }
extern "C" void hlala_seed_batch_free(hlala_seed_batch* S) { delete S; }
