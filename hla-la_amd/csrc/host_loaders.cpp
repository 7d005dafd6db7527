// host_loaders.cpp -- PRG/graph.txt -> hlala_graph_desc, and a binary cache of it.
//
// Text format = Graph::writeToFile / Graph::readFromFile (Graph/Graph.cpp:2225-2327 / :2329-2545) with the per-locus allele code of
// LocusCodeAllocation (Graph/LocusCodeAllocation.cpp:264-312, deCode :32-48):
//     CODE:    <locus>|||<allele>|||<code 0..250>
//     NODES:   <index>|||<level>|||<terminal>
//     EDGES:   <index>|||<locus>|||<count>|||<coded emission byte>|||<from index>|||<to index>[|||<label>|||<pgf_protect>]
// The canonical order of nodes and edges is their order in the file (= the order the reference creates the objects in); node indices
// of the file are arbitrary integers and are renumbered 0.. in line order.  An emission byte equal to '|' appears as "|||||||" and is
// handled the way the reference does (:2338-2365, :2450-2453).
#include <algorithm>
#include <atomic>
#include <exception>
#include <thread>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iterator>
#include <map>
#include <memory>
#include <sstream>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/hlala_gpu.h"

struct hlala_graph_file {
    std::vector<int32_t> node_level, edge_from, edge_to;
    std::vector<uint8_t> edge_label;
    int32_t n_levels = 0;
};

namespace {

thread_local std::string g_loader_error;

const std::string SEP = "|||";                               // separatorForSerialization, Graph.cpp:26

// boost::iter_split(first_finder(SEP)): split at every occurrence of SEP, scanning left to right
void split_sep(const std::string& line, std::vector<std::string>& out)
{
    out.clear();
    size_t p = 0;
    for(;;) {
        size_t q = line.find(SEP, p);
        if(q == std::string::npos) { out.push_back(line.substr(p)); return; }
        out.push_back(line.substr(p, q - p));
        p = q + SEP.size();
    }
}

bool parse_int(const std::string& s, long long& v)
{
    if(s.empty()) return false;
    char* end = nullptr;
    v = strtoll(s.c_str(), &end, 10);
    return end && *end == 0;
}

int fail(const std::string& m) { g_loader_error = m; return HLALA_E_ARG; }

}  // namespace

extern "C" const char* hlala_loader_last_error() { return g_loader_error.c_str(); }

extern "C" int hlala_graph_load_text(const char* path, hlala_graph_file** out)
try {
    if(!path || !out) return HLALA_E_ARG;
    *out = nullptr;
    std::ifstream in(path);
    if(!in.is_open()) return fail(std::string("Cannot open graph file: ") + path);
    const std::string problematic_part = "|||||||", subsitute_problem = "|||SLASH|||", substitue_indicator = "SLASH";
    std::map<std::string, std::map<int, std::string>> coded_values_rev;          // locus -> code -> allele
    std::unordered_map<long long, int32_t> idx2Node;
    hlala_graph_file* g = new hlala_graph_file();
    auto bad = [&](const std::string& m) { delete g; return fail(m); };
    int mode = -1; std::string line; std::vector<std::string> fields;
    int maxLevel = -1; long long lineNo = 0;
    while(std::getline(in, line)) {
        lineNo++;
        while(!line.empty() && (line.back() == '\n' || line.back() == '\r')) line.pop_back();       // Utilities::eraseNL
        if(line.empty()) continue;
        { size_t q = line.find(problematic_part); if(q != std::string::npos) line.replace(q, problematic_part.size(), subsitute_problem); }
        if(line == "CODE:") { mode = 1; continue; }
        if(line == "NODES:") { mode = 2; continue; }
        if(line == "EDGES:") { mode = 3; continue; }
        if(mode < 0) return bad("graph file: data before the first section header (line " + std::to_string(lineNo) + ")");
        split_sep(line, fields);
        if(mode == 1) {
            if(fields.size() != 3) return bad("Cannot read CODE from line, expect 3 fields! Line: " + line);
            long long code; if(!parse_int(fields[2], code) || code < 0 || code > 250) return bad("Weird codedChar value: cannot convert back! " + fields[2]);
            coded_values_rev[fields[0]][(int)code] = fields[1];
        } else if(mode == 2) {
            if(fields.size() != 3) return bad("Cannot node-parse this line (expect 3 fields): " + line);
            long long idx, level; if(!parse_int(fields[0], idx) || !parse_int(fields[1], level) || level < 0 || level > 0x7FFFFFF0LL) return bad("Cannot node-parse this line: " + line);
            if(idx2Node.count(idx)) return bad("node index appears twice: " + fields[0]);
            idx2Node[idx] = (int32_t)g->node_level.size();
            g->node_level.push_back((int32_t)level);
            if(level > maxLevel) maxLevel = (int)level;
        } else {
            if(fields.size() != 6 && fields.size() != 8) return bad("Cannot edge-parse this line (expect 6/8 fields): " + line);
            if(fields[3] == substitue_indicator) fields[3] = "|";
            if(fields[3].size() != 1) return bad("Cannot cast to unsigned char: " + fields[3] + "--" + line);          // lexical_cast<unsigned char>
            const int emission = (unsigned char)fields[3][0];
            long long from_idx, to_idx; if(!parse_int(fields[4], from_idx) || !parse_int(fields[5], to_idx)) return bad("Cannot edge-parse this line: " + line);
            auto lc = coded_values_rev.find(fields[1]);
            if(lc == coded_values_rev.end()) return bad("Non-assigned locus " + fields[1]);
            auto ac = lc->second.find(emission);
            if(ac == lc->second.end()) return bad("Non-assigned allele for " + fields[1] + " value: " + std::to_string(emission));
            if(ac->second.size() != 1) return bad("decoded emission is not a single character for locus " + fields[1]);   // assert(emissionString.length() == 1), :2515
            auto nf = idx2Node.find(from_idx), nt = idx2Node.find(to_idx);
            if(nf == idx2Node.end() || nt == idx2Node.end()) return bad("Edge refers to an unknown node index. Edge line: " + line);
            g->edge_from.push_back(nf->second); g->edge_to.push_back(nt->second); g->edge_label.push_back((uint8_t)ac->second[0]);
        }
    }
    if(g->node_level.empty()) return bad("graph file holds no nodes");
    g->n_levels = maxLevel + 1;
    *out = g;
    return HLALA_OK;
} catch(const std::exception& e_) { g_loader_error = std::string("hlala_graph_load_text: ") + e_.what(); return HLALA_E_ARG; }

extern "C" int hlala_graph_file_desc(const hlala_graph_file* g, hlala_graph_desc* d)
try {
    if(!g || !d) return HLALA_E_ARG;
    d->n_levels = g->n_levels; d->n_nodes = (int32_t)g->node_level.size(); d->n_edges = (int32_t)g->edge_from.size();
    d->node_level = g->node_level.data(); d->edge_from = g->edge_from.data(); d->edge_to = g->edge_to.data(); d->edge_label = g->edge_label.data();
    return HLALA_OK;
} catch(const std::exception& e_) { g_loader_error = std::string("hlala_graph_file_desc: ") + e_.what(); return HLALA_E_ARG; }

extern "C" void hlala_graph_file_free(hlala_graph_file* g) { delete g; }

// ---- binary cache: header {magic, n_levels, n_nodes, n_edges}, then node_level, edge_from, edge_to (int32) and edge_label (bytes)
static const char CACHE_MAGIC[8] = {'H', 'L', 'A', 'L', 'A', 'G', 'R', '1'};

extern "C" int hlala_graph_cache_save(const hlala_graph_desc* d, const char* path)
try {
    if(!d || !path) return HLALA_E_ARG;
    FILE* f = fopen(path, "wb");
    if(!f) return fail(std::string("Cannot open cache file for writing: ") + path);
    int32_t hdr[3] = {d->n_levels, d->n_nodes, d->n_edges};
    bool ok = fwrite(CACHE_MAGIC, 1, 8, f) == 8 && fwrite(hdr, 4, 3, f) == 3 &&
              fwrite(d->node_level, 4, (size_t)d->n_nodes, f) == (size_t)d->n_nodes && fwrite(d->edge_from, 4, (size_t)d->n_edges, f) == (size_t)d->n_edges &&
              fwrite(d->edge_to, 4, (size_t)d->n_edges, f) == (size_t)d->n_edges && fwrite(d->edge_label, 1, (size_t)d->n_edges, f) == (size_t)d->n_edges;
    ok = (fclose(f) == 0) && ok;
    return ok ? HLALA_OK : fail(std::string("short write to ") + path);
} catch(const std::exception& e_) { g_loader_error = std::string("hlala_graph_cache_save: ") + e_.what(); return HLALA_E_ARG; }

extern "C" int hlala_graph_cache_load(const char* path, hlala_graph_file** out)
try {
    if(!path || !out) return HLALA_E_ARG;
    *out = nullptr;
    FILE* f = fopen(path, "rb");
    if(!f) return fail(std::string("Cannot open cache file: ") + path);
    char magic[8]; int32_t hdr[3];
    if(fread(magic, 1, 8, f) != 8 || memcmp(magic, CACHE_MAGIC, 8) != 0 || fread(hdr, 4, 3, f) != 3 || hdr[0] < 1 || hdr[1] < 1 || hdr[2] < 0) { fclose(f); return fail(std::string("not a graph cache of this library: ") + path); }
    hlala_graph_file* g = new hlala_graph_file();
    g->n_levels = hdr[0]; g->node_level.resize((size_t)hdr[1]); g->edge_from.resize((size_t)hdr[2]); g->edge_to.resize((size_t)hdr[2]); g->edge_label.resize((size_t)hdr[2]);
    bool ok = fread(g->node_level.data(), 4, (size_t)hdr[1], f) == (size_t)hdr[1] && fread(g->edge_from.data(), 4, (size_t)hdr[2], f) == (size_t)hdr[2] &&
              fread(g->edge_to.data(), 4, (size_t)hdr[2], f) == (size_t)hdr[2] && fread(g->edge_label.data(), 1, (size_t)hdr[2], f) == (size_t)hdr[2];
    fclose(f);
    if(!ok) { delete g; return fail(std::string("truncated graph cache: ") + path); }
    *out = g;
    return HLALA_OK;
} catch(const std::exception& e_) { g_loader_error = std::string("hlala_graph_cache_load: ") + e_.what(); return HLALA_E_ARG; }

// ------------------------------------------------------------------------------------------------------------------------------
// Reference contigs of a graph directory: sequences.txt + the reference FASTA + translation/<SequenceID>.txt
// (processBAM::initBAM mapper/processBAM.cpp:1183-1402, the constructor :69-88, processBAM::_loadMapping :4389-4457).
//
// One contig per row of sequences.txt = one "interesting interval": its bases are the stretch [Start_1based, Stop_1based] of the
// BAM reference the row names (column Chr, or PRG_<SequenceID> when Chr is empty; PRG-only mode: the whole sequence), its levels are
// the lines of translation/<SequenceID>.txt.  The FASTA is streamed and only the named sequences are kept (the extended reference is
// the whole genome).  Two quirks of the reference are kept because results can depend on them:
//   * the translation file is read with `while(good) { getline; push(StrtoI(line)) }`, so a file ending in a newline yields one extra
//     entry 0 (StrtoI("") == 0): level 0 then "lies under" position <length> of that sequence in graphLevel_2_underlyingSequencePositions.
//     The contig gets one extra position (base 'N', level 0) so that hlala_create builds the same table;
//   * PRG-only mode defines PRG_5 as "N" (:87-88).
namespace {
struct SeqRow { int id; std::string ref; bool hasRange; int start1, stop1; };
}
struct hlala_contigs_file {
    std::vector<int64_t> off; std::vector<uint8_t> seq; std::vector<int32_t> level, seqid;
    std::vector<std::string> refName; std::vector<int32_t> start0, stop0;
    // between hlala_contigs_open_dir and hlala_contigs_load_translations: the rows of sequences.txt and the reference sequences they name
    bool complete = false; std::string dir; std::vector<SeqRow> rows; std::map<std::string, std::string> seqs;
};

namespace {

bool read_wanted_fasta(const std::string& path, std::map<std::string, std::string>& wanted)
{
    std::ifstream f(path.c_str());
    if(!f.is_open()) return false;
    std::string line; std::string* cur = nullptr;
    while(f.good()) {
        std::getline(f, line);
        while(!line.empty() && (line.back() == '\r' || line.back() == '\n')) line.pop_back();
        if(!line.empty() && line[0] == '>') {
            std::string ident = line.substr(1);                                    // readFASTA(file, false): up to the first blank, Utilities.cpp:782-797
            const size_t sp = ident.find(' '); if(sp != std::string::npos) ident = ident.substr(0, sp);
            auto it = wanted.find(ident);
            cur = it == wanted.end() ? nullptr : &it->second;
            if(cur) cur->assign("\x01");                                            // marks "seen" (removed below): a sequence may be empty
        } else if(cur) *cur += line;
    }
    return true;
}
}  // namespace

extern "C" int hlala_contigs_open_dir(const char* graph_dir, int32_t extended_reference_genome, hlala_contigs_file** out)
try {
    if(!graph_dir || !out) return HLALA_E_ARG;
    const std::string dir(graph_dir);
    std::ifstream sf((dir + "/sequences.txt").c_str());
    if(!sf.is_open()) return fail("cannot open " + dir + "/sequences.txt");
    auto chomp = [](std::string& s) { while(!s.empty() && (s.back() == '\r' || s.back() == '\n')) s.pop_back(); };
    auto split_tab = [](const std::string& s) { std::vector<std::string> v; size_t a = 0, p; if(s.empty()) return v; while((p = s.find('\t', a)) != std::string::npos) { v.push_back(s.substr(a, p - a)); a = p + 1; } v.push_back(s.substr(a)); return v; };
    std::string header; std::getline(sf, header); chomp(header);
    const std::vector<std::string> hf = split_tab(header);
    int cId = -1, cChr = -1, cStart = -1, cStop = -1;
    for(size_t i = 0; i < hf.size(); i++) { if(hf[i] == "SequenceID") cId = (int)i; else if(hf[i] == "Chr") cChr = (int)i; else if(hf[i] == "Start_1based") cStart = (int)i; else if(hf[i] == "Stop_1based") cStop = (int)i; }
    if(cId < 0 || cChr < 0 || (extended_reference_genome && (cStart < 0 || cStop < 0))) return fail(dir + "/sequences.txt: columns SequenceID, Chr, Start_1based, Stop_1based expected");
    std::vector<SeqRow> rows; std::string line;
    while(sf.good()) {
        std::getline(sf, line); chomp(line);
        if(line.empty()) continue;
        const std::vector<std::string> lf = split_tab(line);
        if(lf.size() != hf.size()) return fail(dir + "/sequences.txt: a row has " + std::to_string(lf.size()) + " fields, the header " + std::to_string(hf.size()));
        SeqRow r; r.id = atoi(lf[cId].c_str());
        const bool chr = !lf[cChr].empty();
        r.ref = chr ? lf[cChr] : "PRG_" + lf[cId];
        r.hasRange = chr && extended_reference_genome;
        r.start1 = r.hasRange ? atoi(lf[cStart].c_str()) : 1; r.stop1 = r.hasRange ? atoi(lf[cStop].c_str()) : -1;
        rows.push_back(r);
    }
    if(rows.empty()) return fail(dir + "/sequences.txt: no sequences");
    std::string fasta;
    if(extended_reference_genome) {
        std::ifstream pf((dir + "/extendedReferenceGenomePath.txt").c_str());
        if(pf.is_open()) { std::getline(pf, fasta); chomp(fasta); }                                      // Utilities::getFirstLine
        else fasta = dir + "/extendedReferenceGenome/extendedReferenceGenome.fa";
    } else fasta = dir + "/mapping_PRGonly/referenceGenome.fa";
    std::map<std::string, std::string> seqs;
    for(const SeqRow& r : rows) seqs[r.ref] = "";
    if(!read_wanted_fasta(fasta, seqs)) return fail("readFASTA(): Cannot open file " + fasta);
    if(!extended_reference_genome) { auto it = seqs.find("PRG_5"); if(it != seqs.end() && it->second.empty()) it->second = "\x01N"; }      // :87-88
    std::unique_ptr<hlala_contigs_file> C(new hlala_contigs_file());
    for(const SeqRow& r : rows) {
        const std::string& s = seqs[r.ref];
        if(s.empty()) return fail(r.ref + " cannot be found in the reference genome " + fasta);
        const long long len = (long long)s.size() - 1;                                                    // without the "seen" mark
        const long long a = r.start1, b = r.hasRange ? r.stop1 : len;
        if(a < 1 || b > len || b < a) return fail("sequences.txt: interval " + std::to_string(a) + "-" + std::to_string(b) + " outside " + r.ref + " (length " + std::to_string(len) + ")");
        C->seqid.push_back(r.id); C->refName.push_back(r.ref); C->start0.push_back((int32_t)(a - 1)); C->stop0.push_back((int32_t)(b - 1));
    }
    C->dir = dir; C->rows.swap(rows); C->seqs.swap(seqs);
    *out = C.release();
    return HLALA_OK;
} catch(const std::exception& e_) { g_loader_error = std::string("hlala_contigs_open_dir: ") + e_.what(); return HLALA_E_ARG; }

// the translation tables (tens of millions of lines for the MHC graph: the bulk of the directory's reading time): the files are parsed side by side on a few
// threads and appended in the order of sequences.txt
extern "C" int hlala_contigs_load_translations(hlala_contigs_file* C)
try {
    if(!C) return HLALA_E_ARG;
    if(C->complete) return HLALA_OK;
    const std::string& dir = C->dir; const std::vector<SeqRow>& rows = C->rows;
    std::vector<std::vector<int32_t>> levels(rows.size()); std::vector<std::string> errs(rows.size());
    auto parse_one = [&](size_t ri) {
        const SeqRow& r = rows[ri];
        const std::string tf = dir + "/translation/" + std::to_string(r.id) + ".txt";
        // one level per line, read as the reference's loop does (mapper/processBAM.cpp:4406-4412: getline while good(), Utilities::StrtoI = stringstream >> int,
        // Utilities.cpp:644-650): a file that ends in a newline yields one more, empty line = level 0; a line without a leading integer yields 0.  Parsed by hand:
        // the MHC graph has tens of millions of these lines.
        std::vector<int32_t>& lv = levels[ri];
        std::ifstream ts(tf.c_str(), std::ios::binary);
        if(!ts.is_open()) { errs[ri] = "Expected coordinate translation file not found: " + tf; return; }
        std::string all((std::istreambuf_iterator<char>(ts)), std::istreambuf_iterator<char>());
        const char* p = all.data(); const char* end = p + all.size();
        for(;;) {
            const char* nl = (const char*)memchr(p, '\n', (size_t)(end - p));
            const char* le = nl ? nl : end;
            const char* q = p;
            while(q < le && (*q == ' ' || *q == '\t' || *q == '\r' || *q == '\v' || *q == '\f')) q++;
            bool neg = false; if(q < le && (*q == '+' || *q == '-')) { neg = *q == '-'; q++; }
            long long v = 0; bool digits = false, over = false;
            while(q < le && *q >= '0' && *q <= '9') { digits = true; v = v * 10 + (*q - '0'); if(v > 4294967296ll) over = true; q++; }
            if(neg) v = -v;
            if(!digits || over || v > 2147483647ll || v < -2147483648ll) v = 0;                       // failed extraction: 0
            lv.push_back((int32_t)v);
            if(!nl) break;
            p = nl + 1;
        }
    };
    {
        const size_t T = std::min<size_t>(rows.size(), 8);
        std::atomic<size_t> next(0); std::vector<std::thread> th; std::vector<std::exception_ptr> ex(T);
        try { for(size_t t = 0; t < T; t++) th.emplace_back([&, t]() { try { for(;;) { const size_t ri = next.fetch_add(1); if(ri >= rows.size()) break; parse_one(ri); } } catch(...) { ex[t] = std::current_exception(); } }); }
        catch(...) { next.store(rows.size()); for(std::thread& x : th) x.join(); throw; }
        for(std::thread& x : th) x.join();
        for(const std::exception_ptr& e : ex) if(e) std::rethrow_exception(e);
    }
    C->off.assign(1, 0); C->seq.clear(); C->level.clear();
    for(size_t ri = 0; ri < rows.size(); ri++) {
        if(!errs[ri].empty()) return fail(errs[ri]);
        const SeqRow& r = rows[ri];
        const std::string& s = C->seqs[r.ref];
        const std::vector<int32_t>& lv = levels[ri];
        const std::string tf = dir + "/translation/" + std::to_string(r.id) + ".txt";
        const long long a = (long long)C->start0[ri] + 1, b = (long long)C->stop0[ri] + 1;
        const long long n = b - a + 1;
        if((long long)lv.size() < n) return fail(tf + ": " + std::to_string(lv.size()) + " levels for an interval of " + std::to_string(n) + " bases");
        C->seq.insert(C->seq.end(), s.begin() + a, s.begin() + b + 1);                                    // s[0] is the mark: 1-based start a = index a
        C->seq.insert(C->seq.end(), lv.size() - (size_t)n, (uint8_t)'N');
        C->level.insert(C->level.end(), lv.begin(), lv.end());
        C->off.push_back((int64_t)C->seq.size());
        std::vector<int32_t>().swap(levels[ri]);
    }
    C->complete = true; std::vector<SeqRow>().swap(C->rows); std::map<std::string, std::string>().swap(C->seqs);
    return HLALA_OK;
} catch(const std::exception& e_) { g_loader_error = std::string("hlala_contigs_load_translations: ") + e_.what(); return HLALA_E_ARG; }

extern "C" int hlala_contigs_load_dir(const char* graph_dir, int32_t extended_reference_genome, hlala_contigs_file** out)
{
    if(!out) return HLALA_E_ARG;
    hlala_contigs_file* C = nullptr;
    int rc = hlala_contigs_open_dir(graph_dir, extended_reference_genome, &C);
    if(rc != HLALA_OK) return rc;
    rc = hlala_contigs_load_translations(C);
    if(rc != HLALA_OK) { delete C; return rc; }
    *out = C;
    return HLALA_OK;
}
extern "C" int hlala_contigs_file_desc(const hlala_contigs_file* c, hlala_contigs_desc* d)
try {
    if(!c || !d) return HLALA_E_ARG;
    if(!c->complete) { g_loader_error = "hlala_contigs_file_desc before hlala_contigs_load_translations"; return HLALA_E_STATE; }
    d->n_contigs = (int32_t)c->seqid.size(); d->contig_off = c->off.data(); d->contig_seq = c->seq.data(); d->contig_level = c->level.data(); d->contig_seqid = c->seqid.data();
    return HLALA_OK;
} catch(const std::exception& e_) { g_loader_error = std::string("hlala_contigs_file_desc: ") + e_.what(); return HLALA_E_ARG; }
extern "C" int32_t hlala_contigs_file_intervals(const hlala_contigs_file* c, hlala_bam_interval* out, int32_t cap)
{
    if(!c) return -1;
    const int32_t n = (int32_t)c->seqid.size();
    for(int32_t i = 0; out && i < n && i < cap; i++) { out[i].ref_name = c->refName[i].c_str(); out[i].start_0based = c->start0[i]; out[i].stop_0based = c->stop0[i]; out[i].contig = i; }
    return n;
}
extern "C" void hlala_contigs_file_free(hlala_contigs_file* c) { delete c; }
