// host_typer.cpp -- the HLATyper host side around the kernels: which graph levels are which exon column, the allele
// clusters of a locus, G groups, the pile-up and the per-locus result files of HLATyper::HLATypeInference.
//
//   hlala_typer_open        graph loci = column names of the segment files in PRG/segments.txt order (Graph::readGraphLoci,
//                           Graph/Graph.cpp:2563-2614), gene level boundaries (hla/HLATyper.cpp:104-214), the file list of PRG/
//   hlala_typer_locus       exon files of one locus -> combined exon columns, allele sequences, clusters of identical alleles
//                           (hla/HLATyper.cpp:1180-1372; find_file_for_exon :3130-3200; the exon table fill_loci_2_exons :2812-2846)
//   hlala_locus_write_files pile-up, read IDs, all-pairs table, column incompatibilities, best-guess rows (:1883-2044, :2451-2488,
//                           :2543-2759; G groups :4086-4207)
//
// Text goes through std::ostream with default formatting, exactly the calls the reference makes (Utilities::DtoStr / ItoStr are
// `stringstream << value`, Utilities.cpp:576-597), so numbers print the same way.  Data is flat: positions are bucketed by exon
// column with a counting pass, never as maps of maps of structs.
#include <dirent.h>
#include <sys/stat.h>

#include <algorithm>
#include <mutex>
#include <exception>
#include <string_view>
#include <functional>
#include <thread>
#include <type_traits>
#include <atomic>
#include <charconv>
#include <cmath>
#include <cstring>
#include <fstream>
#include <map>
#include <memory>
#include <set>
#include <sstream>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/hlala_gpu.h"
#include "host_internal.h"

namespace {

thread_local std::string g_err;
int fail(int code, const std::string& m) { g_err = m; return code; }

// Utilities::split(string, string), Utilities.cpp:531-568 (empty fields are kept)
std::vector<std::string> split(const std::string& in, const std::string& d)
{
    std::vector<std::string> out;
    if(in.empty()) return out;
    size_t s = 0, p;
    while((p = in.find(d, s)) != std::string::npos) { out.push_back(in.substr(s, p - s)); s = p + d.size(); }
    out.push_back(in.substr(s));
    return out;
}
void erase_nl(std::string& s) { while(!s.empty() && (s.back() == '\r' || s.back() == '\n')) s.pop_back(); }
std::string join(const std::vector<std::string>& v, const std::string& d)
{
    std::string r;
    for(size_t i = 0; i < v.size(); i++) { if(i) r += d; r += v[i]; }
    return r;
}
// ItoStr / DtoStr: what `stream << v` prints.  Integers and finite floating-point values go through std::to_chars -- for a double the stream's default format is
// %g with six significant digits = std::chars_format::general, precision 6 (checked against operator<< on 15 M values) -- because a std::stringstream per
// number was most of the time of the pile-up files; everything else (characters, bool, non-finite values) takes the stream.
template <class T> typename std::enable_if<!std::is_arithmetic<T>::value || std::is_same<T, bool>::value || std::is_same<T, char>::value || std::is_same<T, signed char>::value || std::is_same<T, unsigned char>::value, std::string>::type
to_str(T v) { std::stringstream s; s << v; return s.str(); }
template <class T> typename std::enable_if<std::is_integral<T>::value && !std::is_same<T, bool>::value && !std::is_same<T, char>::value && !std::is_same<T, signed char>::value && !std::is_same<T, unsigned char>::value, std::string>::type
to_str(T v) { char b[24]; const auto r = std::to_chars(b, b + sizeof b, v); return std::string(b, r.ptr); }
template <class T> typename std::enable_if<std::is_floating_point<T>::value, std::string>::type
to_str(T v)
{
    if(!std::isfinite(v)) { std::stringstream s; s << v; return s.str(); }
    char b[48]; const auto r = std::to_chars(b, b + sizeof b, v, std::chars_format::general, 6); return std::string(b, r.ptr);
}

// the reference's reading loop: `while(stream.good()) { getline; eraseNL; push }` -- a final empty line is part of the result
bool read_lines(const std::string& path, std::vector<std::string>& lines)
{
    std::ifstream f(path.c_str());
    if(!f.is_open()) return false;
    while(f.good()) { std::string l; std::getline(f, l); erase_nl(l); lines.push_back(l); }
    return true;
}

// Utilities::PhredToPCorrect, Utilities.cpp:357-377
double phred_to_pcorrect(unsigned char q)
{
    if(q == 0) return -1;
    return 1 - exp(log(10.0) * (((double)q - 33) / -10.0));
}

// HLATyper::simpleChiSq for two classes (hla/HLATyper.cpp:4258-4320): 1 - cdf of chi-squared with ONE degree of freedom.  boost
// evaluates that cdf as gamma_p(1/2, x/2) = 1 - erfc(sqrt(x/2)) for half-integer shape; the same two subtractions are made here
double chi_sq_p(const double observed[2], const double expected[2])
{
    double statistic = 0;
    for(int i = 0; i < 2; i++) statistic += pow((observed[i] - expected[i]), 2) / expected[i];
    const double cdf = 1 - erfc(sqrt(statistic / 2));
    return 1 - cdf;
}

}  // namespace

// graphLocus_2_levels: level of a name.  An open-addressing table of level numbers over the names themselves (no node, no second copy of five million
// strings: building a std::unordered_map<std::string, int> of the MHC graph's level names took four seconds, longer than the BAM decoder runs); filled from
// several threads (a slot is claimed by compare-and-swap), read-only afterwards.
struct LevelIndex {
    const std::vector<std::string>* names = nullptr;
    std::unique_ptr<std::atomic<int32_t>[]> slot; size_t mask = 0;
    static size_t hash(const char* p, size_t n) { return std::hash<std::string_view>()(std::string_view(p, n)); }
    void reset(const std::vector<std::string>* nm)
    {
        names = nm; size_t cap = 16; while(cap < 2 * nm->size() + 2) cap <<= 1;
        slot.reset(new std::atomic<int32_t>[cap]); mask = cap - 1;
        for(size_t i = 0; i < cap; i++) slot[i].store(-1, std::memory_order_relaxed);
    }
    // false: another level carries the same name
    bool insert(int32_t level)
    {
        const std::string& s = (*names)[(size_t)level];
        for(size_t h = hash(s.data(), s.size()) & mask;; h = (h + 1) & mask) {
            int32_t cur = slot[h].load(std::memory_order_acquire);
            if(cur == -1) { if(slot[h].compare_exchange_strong(cur, level, std::memory_order_acq_rel)) return true; }
            if((*names)[(size_t)cur] == s) return false;
        }
    }
    int find(const char* p, size_t n) const
    {
        if(!names || !slot) return -1;
        for(size_t h = hash(p, n) & mask;; h = (h + 1) & mask) {
            const int32_t cur = slot[h].load(std::memory_order_relaxed);
            if(cur == -1) return -1;
            const std::string& s = (*names)[(size_t)cur];
            if(s.size() == n && memcmp(s.data(), p, n) == 0) return cur;
        }
    }
    int find(const std::string& s) const { return find(s.data(), s.size()); }
};

struct hlala_typer {
    std::string graphDir;
    std::vector<std::string> levelNames;                       // graphLoci
    LevelIndex levelOf;                                        // graphLocus_2_levels
    std::vector<std::string> files;                            // files_in_graphDir
    std::vector<std::string> geneNames; std::vector<int> geneFirst, geneLast;       // graphgene_levelBoundaries in std::map (name) order
    std::map<std::string, std::string> alleleToG; std::set<std::string> gLoci;      // read_G_alleles
};

struct hlala_locus {
    std::string name;
    int C = 0, P = 0, levelMin = -1, levelMax = -1, nTypes = 0;
    std::vector<uint8_t> clusterSeq;                            // [C*P]
    std::vector<std::vector<std::string>> members;              // HLAtype_clusters (std::set order)
    std::vector<std::string> clusterId;                         // members joined by ";"
    std::map<std::string, int> typeCluster;                     // HLAtype_2_clusterID
    std::vector<int> colLevel, colExon, colExonPos, exonLength, levelToExon;
    const hlala_typer* typer = nullptr;
};

extern "C" const char* hlala_typer_last_error(void) { return g_err.c_str(); }

extern "C" int hlala_typer_open(const char* graph_dir, hlala_typer** out)
try {
    if(!graph_dir || !out) return fail(HLALA_E_ARG, "hlala_typer_open: null argument");
    std::unique_ptr<hlala_typer> T(new hlala_typer());
    T->graphDir = graph_dir;
    const std::string prg = T->graphDir + "/PRG";
    std::vector<std::string> segLines;
    if(!read_lines(prg + "/segments.txt", segLines)) return fail(HLALA_E_ARG, "Cannot open segments file: " + prg + "/segments.txt");
    // graph loci: the column names of every segment file, in file order (Graph::readGraphLoci).  The first lines are read and cut into names side by side
    // (millions of names for the MHC graph), every segment's names land at the segment's place in the list of levels.
    const size_t nSeg = segLines.size();
    std::vector<std::vector<std::string>> firstFields(nSeg); std::vector<std::string> segErr(nSeg);
    auto in_parallel = [&](size_t n, const std::function<void(size_t)>& fn) {
        const size_t Tn = std::max<size_t>(1, std::min<size_t>(std::min<size_t>(n, 16), (size_t)hlala_host::host_cpu_budget()));
        std::atomic<size_t> next(0); std::vector<std::exception_ptr> ex(Tn); std::vector<std::thread> th;
        auto work = [&](size_t t) { try { for(;;) { const size_t i = next.fetch_add(1); if(i >= n) break; fn(i); } } catch(...) { ex[t] = std::current_exception(); } };
        try { for(size_t t = 1; t < Tn; t++) th.emplace_back(work, t); } catch(...) { next.store(n); for(std::thread& x : th) x.join(); throw; }
        work(0);
        for(std::thread& x : th) x.join();
        for(const std::exception_ptr& e : ex) if(e) std::rethrow_exception(e);
    };
    in_parallel(nSeg, [&](size_t li) {
        const std::string& l = segLines[li];
        if(l.empty()) return;
        std::ifstream f((prg + "/" + l).c_str());
        if(!f.is_open()) { segErr[li] = "Cannot open one segment file: " + prg + "/" + l; return; }
        std::string first; std::getline(f, first); erase_nl(first);
        firstFields[li] = split(first, " ");
    });
    for(const std::string& e : segErr) if(!e.empty()) return fail(HLALA_E_ARG, e);
    std::vector<size_t> base(nSeg + 1, 0);
    for(size_t li = 0; li < nSeg; li++) base[li + 1] = base[li] + (firstFields[li].size() > 1 ? firstFields[li].size() - 1 : 0);
    if(base[nSeg] > 0x7FFFFFF0ull) return fail(HLALA_E_CAPACITY, "more than 2^31 graph levels");
    T->levelNames.resize(base[nSeg]);
    std::vector<std::string> keepFirst(nSeg);                                 // the first field of every segment file ("IndividualID" for a gene segment)
    in_parallel(nSeg, [&](size_t li) {
        std::vector<std::string>& ff = firstFields[li];
        if(!ff.empty()) keepFirst[li] = ff[0];
        for(size_t i = 1; i < ff.size(); i++) T->levelNames[base[li] + i - 1] = std::move(ff[i]);
        std::vector<std::string>().swap(ff);
    });
    T->levelOf.reset(&T->levelNames);
    std::vector<long long> dup(nSeg, -1);
    in_parallel(nSeg, [&](size_t li) { for(size_t k = base[li]; k < base[li + 1]; k++) if(!T->levelOf.insert((int32_t)k)) { dup[li] = (long long)k; return; } });
    for(size_t li = 0; li < nSeg; li++) if(dup[li] >= 0) return fail(HLALA_E_ARG, "graph locus appears twice: " + T->levelNames[(size_t)dup[li]]);           // assert, hla/HLATyper.cpp:90
    // gene boundaries, hla/HLATyper.cpp:104-214 (the level of a segment's i-th name is its place in the list: the names are distinct)
    std::map<std::string, std::pair<int, int>> bounds;
    for(size_t li = 0; li < nSeg; li++) {
        const std::string& l = segLines[li];
        if(l.empty()) continue;
        const std::vector<std::string> us = split(l, "_");
        if(us.size() < 2) return fail(HLALA_E_ARG, "segments.txt: unexpected entry " + l);
        if(us[1] != "gene") continue;
        if(us.size() < 5) return fail(HLALA_E_ARG, "segments.txt: unexpected gene entry " + l);
        const std::string& gene = us[2];
        if(keepFirst[li] != "IndividualID") return fail(HLALA_E_ARG, "segment file without IndividualID header: " + l);
        if(!bounds.count(gene)) bounds[gene] = std::make_pair(-1, -1);
        if(base[li + 1] > base[li]) {
            std::pair<int, int>& bd = bounds[gene];
            const int lo = (int)base[li], hi = (int)base[li + 1] - 1;
            if(bd.first == -1 || lo < bd.first) bd.first = lo;
            if(bd.second == -1 || hi > bd.second) bd.second = hi;
        }
    }
    for(const auto& g : bounds) { T->geneNames.push_back(g.first); T->geneFirst.push_back(g.second.first); T->geneLast.push_back(g.second.second); }
    // Utilities::filesInDirectory, Utilities.cpp:1337-1366
    DIR* d = opendir(prg.c_str());
    if(!d) return fail(HLALA_E_ARG, "Cannot open directory " + prg);
    while(dirent* e = readdir(d)) { std::string n(e->d_name); if(n == "." || n == "..") continue; T->files.push_back(prg + "/" + n); }
    closedir(d);
    *out = T.release();
    return HLALA_OK;
} catch(const std::exception& e_) { g_err = std::string("hlala_typer_open: ") + e_.what(); return HLALA_E_ARG; }
extern "C" void hlala_typer_close(hlala_typer* t) { delete t; }
extern "C" int32_t hlala_typer_n_levels(const hlala_typer* t) { return t ? (int32_t)t->levelNames.size() : -1; }
extern "C" const char* hlala_typer_level_name(const hlala_typer* t, int32_t level) { return (t && level >= 0 && level < (int)t->levelNames.size()) ? t->levelNames[level].c_str() : nullptr; }
extern "C" int32_t hlala_typer_level_of(const hlala_typer* t, const char* id) { if(!t || !id) return -1; return t->levelOf.find(id, strlen(id)); }
extern "C" int32_t hlala_typer_n_genes(const hlala_typer* t) { return t ? (int32_t)t->geneNames.size() : -1; }
extern "C" int hlala_typer_gene(const hlala_typer* t, int32_t i, const char** name, int32_t* first_level, int32_t* last_level)
try {
    if(!t || i < 0 || i >= (int)t->geneNames.size()) return HLALA_E_ARG;
    if(name) *name = t->geneNames[i].c_str();
    if(first_level) *first_level = t->geneFirst[i];
    if(last_level) *last_level = t->geneLast[i];
    return HLALA_OK;
} catch(const std::exception& e_) { g_err = std::string("hlala_typer_gene: ") + e_.what(); return HLALA_E_ARG; }

// ---- G groups, hla/HLATyper.cpp:4150-4207
extern "C" int hlala_typer_load_g_groups(hlala_typer* t, const char* path)
try {
    if(!t || !path) return fail(HLALA_E_ARG, "hlala_typer_load_g_groups: null argument");
    std::vector<std::string> lines;
    if(!read_lines(path, lines)) return fail(HLALA_E_ARG, std::string("Can't open file ") + path);
    t->alleleToG.clear(); t->gLoci.clear();
    for(const std::string& line : lines) {
        if(line.empty() || line[0] == '#') continue;
        const std::vector<std::string> c = split(line, ";");
        if(c.size() < 2 || c[0].empty() || c[0].back() != '*') return fail(HLALA_E_ARG, "G group file: unexpected line " + line);
        const std::string& locusStar = c[0];
        t->gLoci.insert(locusStar.substr(0, locusStar.size() - 1));
        std::string code;
        if(!c.back().empty()) code = c.back();
        else { if(c.size() != 3) return fail(HLALA_E_ARG, "G group file: unexpected line " + line); code = c[1]; }
        code = locusStar + code;
        for(const std::string& a : split(c[1], "/")) t->alleleToG[locusStar + a] = code;
    }
    return HLALA_OK;
} catch(const std::exception& e_) { g_err = std::string("hlala_typer_load_g_groups: ") + e_.what(); return HLALA_E_ARG; }

namespace {
// translate_allele_list_to_G_allele, hla/HLATyper.cpp:4095-4148
std::string to_g_group(const hlala_typer& T, const std::vector<std::string>& alleles, bool& perfectly)
{
    std::map<std::string, int> groups;
    for(const std::string& a : alleles) { auto it = T.alleleToG.find(a); if(it == T.alleleToG.end()) continue; groups[it->second]++; }
    if(groups.empty()) { perfectly = false; return join(alleles, ";"); }
    if(groups.size() == 1) { perfectly = true; return groups.begin()->first; }
    perfectly = false;
    std::vector<std::string> keys;                                                                  // Utilities::get_map_keys_sorted_by_value, Utilities.cpp:1396-1415
    for(const auto& e : groups) keys.push_back(e.first);
    std::sort(keys.begin(), keys.end(), [&](const std::string& l, const std::string& r) { return groups.at(l) < groups.at(r); });
    std::reverse(keys.begin(), keys.end());
    return keys[0];
}

const char* const kExonTable[][3] = {      // fill_loci_2_exons, hla/HLATyper.cpp:2812-2846
    {"A", "exon_2", "exon_3"}, {"B", "exon_2", "exon_3"}, {"C", "exon_2", "exon_3"}, {"DQA1", "exon_2", nullptr}, {"DQB1", "exon_2", nullptr}, {"DRB1", "exon_2", nullptr},
    {"DPA1", "exon_2", nullptr}, {"DPB1", "exon_2", nullptr}, {"DRA", "exon_2", nullptr}, {"DRB3", "exon_2", nullptr}, {"DRB4", "exon_2", nullptr},
    {"E", "exon_2", "exon_3"}, {"F", "exon_2", "exon_3"}, {"G", "exon_2", "exon_3"}, {"H", "exon_2", "exon_3"}, {"J", "exon_2", "exon_3"}, {"K", "exon_2", "exon_3"},
    {"L", "exon_2", "exon_3"}, {"V", "exon_2", "exon_3"}};

// find_file_for_exon, hla/HLATyper.cpp:3130-3200: <n>_gene_<HLA-locus | locus>_<m>_exon_<N>.txt, the last match in directory order
std::string find_exon_file(const hlala_typer& T, const std::string& locus, const std::string& exon, std::string& err)
{
    const std::vector<std::string> parts = split(exon, "_");
    if(parts.size() != 2 || parts[0] != "exon" || atoi(parts[1].c_str()) <= 0) { err = "exon id must look like exon_2: " + exon; return ""; }
    const std::string want = to_str(atoi(parts[1].c_str())) + ".txt";
    std::string found;
    for(const std::string& f : T.files) {
        const std::vector<std::string> sl = split(f, "/");
        if(sl.empty()) continue;
        const std::vector<std::string> us = split(sl.back(), "_");
        if(us.size() >= 6 && us[1] == "gene" && (us[2] == "HLA-" + locus || us[2] == locus) && us[4] == "exon" && us[5] == want) found = f;
    }
    if(found.empty()) err = "no file for locus " + locus + ", " + exon + " in " + T.graphDir + "/PRG";
    return found;
}
}  // namespace

// translate_allele_list_to_G_allele (hla/HLATyper.cpp:4095-4148) for a ';'-joined allele list (the members of a cluster): out receives
// the G group (or the list itself when no member is known), *perfectly whether all known members share it; HLALA_E_STATE when no
// G table is loaded or the locus (text before '*') is not in it (can_translateToG_locus, :4086-4092)
extern "C" int hlala_typer_g_translate(const hlala_typer* t, const char* alleles, char* out, int32_t cap, int32_t* perfectly)
try {
    if(!t || !alleles || !out || cap < 1) return fail(HLALA_E_ARG, "hlala_typer_g_translate: null argument");
    const std::vector<std::string> al = split(alleles, ";");
    if(al.empty() || t->alleleToG.empty()) return fail(HLALA_E_STATE, "hlala_typer_g_translate: no G group table loaded");
    const size_t star = al[0].find('*');
    if(star == std::string::npos || !t->gLoci.count(al[0].substr(0, star))) return fail(HLALA_E_STATE, "hlala_typer_g_translate: locus not in the G group table");
    bool perf = false; const std::string g = to_g_group(*t, al, perf);
    if((int32_t)g.size() + 1 > cap) return fail(HLALA_E_CAPACITY, "hlala_typer_g_translate: output buffer too small");
    memcpy(out, g.c_str(), g.size() + 1);
    if(perfectly) *perfectly = perf ? 1 : 0;
    return HLALA_OK;
} catch(const std::exception& e_) { g_err = std::string("hlala_typer_g_translate: ") + e_.what(); return HLALA_E_ARG; }

extern "C" int hlala_typer_locus(const hlala_typer* t, const char* locus, int32_t n_exons, const char* const* exon_ids, hlala_locus** out)
try {
    if(!t || !locus || !out) return fail(HLALA_E_ARG, "hlala_typer_locus: null argument");
    std::vector<std::string> exons;
    if(exon_ids) for(int i = 0; i < n_exons; i++) exons.push_back(exon_ids[i]);
    else for(const auto& row : kExonTable) if(locus == std::string(row[0])) { exons.push_back(row[1]); if(row[2]) exons.push_back(row[2]); }
    if(exons.empty()) return fail(HLALA_E_ARG, std::string("no exons known for locus ") + locus);        // assert(loci_2_exons.at(locus).size()), :1180
    std::unique_ptr<hlala_locus> L(new hlala_locus());
    L->name = locus; L->typer = t;
    std::map<std::string, std::string> seqOf;                                                           // combined_exon_sequences
    for(size_t exonI = 0; exonI < exons.size(); exonI++) {
        std::string err; const std::string file = find_exon_file(*t, locus, exons[exonI], err);
        if(file.empty()) return fail(HLALA_E_ARG, err);
        std::vector<std::string> lines;
        if(!read_lines(file, lines)) return fail(HLALA_E_ARG, "Can't read file " + file);
        const std::vector<std::string> head = split(lines[0], " ");
        if(head.empty() || head[0] != "IndividualID") return fail(HLALA_E_ARG, file + ": first field must be IndividualID");
        if(head.size() < 2) return fail(HLALA_E_ARG, file + ": no columns");
        const int first = t->levelOf.find(head[1]), last = t->levelOf.find(head.back());
        if(first < 0 || last < 0) return fail(HLALA_E_ARG, file + ": column names are not graph loci");
        if(!(last > first) || (int)head.size() - 1 != last - first + 1) return fail(HLALA_E_ARG, "locus " + L->name + " " + exons[exonI] + " (" + file + "): problem with expected graph length");
        const int len = last - first + 1;
        for(int i = 0; i < len; i++) {
            if(t->levelOf.find(head[(size_t)i + 1]) != first + i) return fail(HLALA_E_ARG, file + ": columns are not consecutive graph levels");       // assert, :1247
            L->colLevel.push_back(first + i); L->colExon.push_back((int)exonI); L->colExonPos.push_back(i);
            if(L->levelMin == -1 || L->levelMin > first + i) L->levelMin = first + i;
            if(L->levelMax == -1 || L->levelMax < first + i) L->levelMax = first + i;
        }
        L->exonLength.push_back(len);
        for(size_t li = 1; li < lines.size(); li++) {
            if(lines[li].empty()) continue;
            const std::vector<std::string> fields = split(lines[li], " ");
            if(fields.size() != head.size()) return fail(HLALA_E_ARG, file + ": line " + to_str(li + 1) + " has " + to_str(fields.size()) + " fields, header has " + to_str(head.size()));
            const std::string& type = fields[0];
            if(type.find(":") == std::string::npos) continue;                                            // :1275
            std::string s; for(size_t i = 1; i < fields.size(); i++) s += fields[i];
            if(exonI == 0) { if(seqOf.count(type)) return fail(HLALA_E_ARG, file + ": allele listed twice: " + type); seqOf[type] = s; }
            else { auto it = seqOf.find(type); if(it == seqOf.end()) return fail(HLALA_E_ARG, file + ": allele missing from the first exon: " + type); it->second += s; }
        }
        if(seqOf.empty()) return fail(HLALA_E_ARG, file + ": no alleles");
    }
    L->P = (int)L->colLevel.size();
    L->levelToExon.assign((size_t)(L->levelMax - L->levelMin + 1), -1);
    for(int pI = 0; pI < L->P; pI++) L->levelToExon[(size_t)(L->colLevel[pI] - L->levelMin)] = pI;      // graphLevel_2_exonPosition (later columns win, as map assignment)
    // clusters of identical sequences in allele-name order, :1322-1372
    std::unordered_map<std::string, int> clusterOfSeq;
    std::vector<std::set<std::string>> clusters;
    size_t sequenceL = 0; bool firstSeq = true;
    for(const auto& ts : seqOf) {
        if(firstSeq) { sequenceL = ts.second.size(); firstSeq = false; }
        else if(sequenceL != ts.second.size()) return fail(HLALA_E_ARG, "locus " + L->name + ": allele " + ts.first + " has a different sequence length");
        auto it = clusterOfSeq.find(ts.second);
        int c;
        if(it != clusterOfSeq.end()) c = it->second;
        else { c = (int)clusters.size(); clusters.emplace_back(); clusterOfSeq[ts.second] = c; L->clusterSeq.insert(L->clusterSeq.end(), ts.second.begin(), ts.second.end()); }
        clusters[c].insert(ts.first); L->typeCluster[ts.first] = c;
    }
    if((int)sequenceL != L->P) return fail(HLALA_E_ARG, "locus " + L->name + ": allele sequences have " + to_str(sequenceL) + " characters, the exons have " + to_str(L->P) + " columns (multi-character fields?)");
    L->C = (int)clusters.size(); L->nTypes = (int)seqOf.size();
    for(const auto& c : clusters) { L->members.emplace_back(c.begin(), c.end()); L->clusterId.push_back(join(L->members.back(), ";")); }
    *out = L.release();
    return HLALA_OK;
} catch(const std::exception& e_) { g_err = std::string("hlala_typer_locus: ") + e_.what(); return HLALA_E_ARG; }
extern "C" void hlala_locus_free(hlala_locus* l) { delete l; }
extern "C" int hlala_locus_get(const hlala_locus* l, hlala_locus_info* o)
try {
    if(!l || !o) return HLALA_E_ARG;
    o->n_clusters = l->C; o->n_columns = l->P; o->n_exons = (int32_t)l->exonLength.size(); o->level_min = l->levelMin; o->level_max = l->levelMax; o->n_types = l->nTypes;
    o->cluster_seq = l->clusterSeq.data(); o->level_to_exon = l->levelToExon.data(); o->col_level = l->colLevel.data(); o->col_exon = l->colExon.data();
    o->col_exon_pos = l->colExonPos.data(); o->exon_length = l->exonLength.data();
    return HLALA_OK;
} catch(const std::exception& e_) { g_err = std::string("hlala_locus_get: ") + e_.what(); return HLALA_E_ARG; }
extern "C" const char* hlala_locus_cluster_id(const hlala_locus* l, int32_t c) { return (l && c >= 0 && c < l->C) ? l->clusterId[c].c_str() : nullptr; }
extern "C" int32_t hlala_locus_type_cluster(const hlala_locus* l, const char* type) { if(!l || !type) return -1; auto it = l->typeCluster.find(type); return it == l->typeCluster.end() ? -1 : it->second; }

// k-mers of one cluster's sequence, exon by exon with gaps removed (calculcatekMerPresence, hla/HLATyper.cpp:2652-2688): n_total counts
// every k-mer, the ones without '*' are written to `queries` (k characters each) for hlala_kmer_presence
extern "C" int hlala_locus_cluster_kmers(const hlala_locus* l, int32_t cluster, int32_t k, char* queries, int32_t cap_queries, int32_t* n_queries, int32_t* n_total)
try {
    if(!l || cluster < 0 || cluster >= l->C || k <= 0 || !n_queries || !n_total) return HLALA_E_ARG;
    int nq = 0, nt = 0; bool overflow = false;
    const uint8_t* s = l->clusterSeq.data() + (size_t)cluster * l->P;
    int col = 0;
    for(size_t e = 0; e < l->exonLength.size(); e++) {
        std::string ex;
        for(int i = 0; i < l->exonLength[e]; i++, col++) if(s[col] != '_') ex.push_back((char)s[col]);
        if((int)ex.size() < k) continue;
        for(size_t i = 0; i + (size_t)k <= ex.size(); i++) {
            nt++;
            if(ex.find('*', i) < i + (size_t)k) continue;
            if(queries && nq < cap_queries) memcpy(queries + (size_t)nq * k, ex.data() + i, (size_t)k); else if(queries) overflow = true;
            nq++;
        }
    }
    *n_queries = nq; *n_total = nt;
    return overflow ? HLALA_E_CAPACITY : HLALA_OK;
} catch(const std::exception& e_) { g_err = std::string("hlala_locus_cluster_kmers: ") + e_.what(); return HLALA_E_ARG; }

// ---- result files ------------------------------------------------------------------------------------------------------------
extern "C" int hlala_typer_begin_output(const char* out_dir, double unaccounted_min_fraction)
try {
    if(!out_dir) return fail(HLALA_E_ARG, "hlala_typer_begin_output: null argument");
    struct stat sb;
    if(stat(out_dir, &sb) != 0 && mkdir(out_dir, 0775) != 0) return fail(HLALA_E_ARG, std::string("cannot create ") + out_dir);
    const std::string field = "NColumns_UnaccountedAllele_fGT" + to_str(unaccounted_min_fraction);
    std::ofstream a((std::string(out_dir) + "/R1_bestguess.txt").c_str()), g((std::string(out_dir) + "/R1_bestguess_G.txt").c_str());
    if(!a.is_open() || !g.is_open()) return fail(HLALA_E_ARG, std::string("cannot write to ") + out_dir);
    a << "Locus" << "\t" << "Chromosome" << "\t" << "Allele" << "\t" << "Q1" << "\t" << "Q2" << "\t" << "AverageCoverage" << "\t" << "CoverageFirstDecile" << "\t" << "MinimumCoverage" << "\t" << "proportionkMersCovered" << "\t" << "LocusAvgColumnError" << "\t" << field << "\n";
    g << "Locus" << "\t" << "Chromosome" << "\t" << "Allele" << "\t" << "Q1" << "\t" << "Q2" << "\t" << "AverageCoverage" << "\t" << "CoverageFirstDecile" << "\t" << "MinimumCoverage" << "\t" << "proportionkMersCovered" << "\t" << "LocusAvgColumnError" << "\t" << field << "\t" << "perfectG" << "\n";
    std::ofstream hst((std::string(out_dir) + "/histogram_matchesPerRead.txt").c_str());
    if(!hst.is_open()) return fail(HLALA_E_ARG, std::string("cannot write to ") + out_dir);
    hst << "Locus" << "\t" << "Level" << "Value" << "\n";                                            // sic, hla/HLATyper.cpp:1145
    return HLALA_OK;
} catch(const std::exception& e_) { g_err = std::string("hlala_typer_begin_output: ") + e_.what(); return HLALA_E_ARG; }
extern "C" int hlala_typer_end_output(const char* out_dir, const char* loci_comma_separated, int32_t very_conservative_read_likelihoods)
try {
    if(!out_dir || !loci_comma_separated) return fail(HLALA_E_ARG, "hlala_typer_end_output: null argument");
    std::ofstream p((std::string(out_dir) + "/R1_parameters.txt").c_str());
    if(!p.is_open()) return fail(HLALA_E_ARG, std::string("cannot write to ") + out_dir);
    p << "Loci" << " = " << loci_comma_separated << "\n";
    p << "veryConservativeReadLikelihoods" << " = " << (very_conservative_read_likelihoods != 0) << "\n";
    return HLALA_OK;
} catch(const std::exception& e_) { g_err = std::string("hlala_typer_end_output: ") + e_.what(); return HLALA_E_ARG; }

// summaryStatistics.txt, hla/HLATyper.cpp:1030-1125
extern "C" int hlala_typer_write_summary(const char* out_dir, int32_t n_units, int32_t unpaired, const uint8_t* unit_mask, const hlala_unit_stats_out* st,
                                         double insert_mean, double insert_sd, int32_t min_alignment_length_unpaired)
try {
    if(!out_dir || !st || n_units < 0) return fail(HLALA_E_ARG, "hlala_typer_write_summary: null argument");
    size_t nPaired = 0, nUnpaired = 0;
    int strandsValid = 0, pairedPerfect = 0, oneReadPerfect = 0, unpairedPerfect = 0, strandsValidDistanceOK = 0, unpairedLongEnough = 0;
    std::vector<double> distances; double pairedSum = 0, unpairedSum = 0;
    for(int32_t u = 0; u < n_units; u++) {
        if((unit_mask && !unit_mask[u]) || !st->valid[u]) continue;
        if(!unpaired) {
            nPaired++;
            if(st->strands_valid[u]) {
                strandsValid++;
                const double d = st->distance[u]; distances.push_back(d);
                if(std::abs(d - insert_mean) <= (5 * insert_sd)) strandsValidDistanceOK++;
            }
            const double f1 = st->fraction_ok[2 * u], f2 = st->fraction_ok[2 * u + 1];
            if(f1 == 1) pairedPerfect++;
            if(f2 == 1) pairedPerfect++;
            if((f1 == 1) || (f2 == 1)) oneReadPerfect++;
            pairedSum += f1; pairedSum += f2;
        } else {
            nUnpaired++;
            const double f1 = st->fraction_ok[2 * u];
            if(st->n_columns[2 * u] >= min_alignment_length_unpaired) unpairedLongEnough++;
            if(f1 == 1) unpairedPerfect++;
            unpairedSum += f1;
        }
    }
    std::sort(distances.begin(), distances.end());                                                     // meanMedian, :3103-3121
    double S = 0; for(double d : distances) S += d;
    const double mean = distances.empty() ? 0 : S / (double)distances.size(), median = distances.empty() ? 0 : distances[distances.size() / 2];
    const double pairedAvg = nPaired > 0 ? (pairedSum / (2.0 * (double)nPaired)) : 0, unpairedAvg = nUnpaired > 0 ? (unpairedSum / ((double)nUnpaired)) : 0;
    auto perc = [](double v1, double v2) { return to_str((v1 / v2) * 100); };                         // printPerc, :3124-3128
    struct stat sb;
    if(stat(out_dir, &sb) != 0 && mkdir(out_dir, 0775) != 0) return fail(HLALA_E_ARG, std::string("cannot create ") + out_dir);
    std::ofstream o((std::string(out_dir) + "/summaryStatistics.txt").c_str());
    if(!o.is_open()) return fail(HLALA_E_ARG, std::string("cannot write to ") + out_dir);
    o << "\nRead alignment statistics:\n";
    o << "\t - Total number (paired) alignments:                 " << nPaired << "\n";
    o << "\t\t - Alignment pairs with strands OK:                  " << strandsValid << " (" << perc(strandsValid, nPaired) << "%)\n";
    o << "\t\t - Alignment pairs with strands OK && distance OK:   " << strandsValidDistanceOK << " (" << perc(strandsValidDistanceOK, nPaired) << "%)\n";
    o << "\t\t - Alignment pairs with strands OK, mean distance:   " << mean << "\n";
    o << "\t\t - Alignment pairs with strands OK, median distance: " << median << "\n";
    o << "\t\t - Alignment pairs, average fraction alignment OK:   " << pairedAvg << "\n";
    o << "\t\t - Alignment pairs, at least one alignment perfect:   " << oneReadPerfect << "\n";
    o << "\t\t - Single alignments, perfect (total):   " << pairedPerfect << " (" << nPaired * 2 << ")\n";
    o << "\t - Total number (unpaired) alignments:                 " << nUnpaired << "\n";
    o << "\t\t - Alignment pairs, average fraction alignment OK:   " << unpairedAvg << "\n";
    o << "\t\t - Single alignments, perfect (total):   " << unpairedPerfect << " (" << nUnpaired * 2 << ")\n";
    o << "\t\t - Alignments with length >= " << min_alignment_length_unpaired << ":   " << unpairedLongEnough << "\n";
    return HLALA_OK;
} catch(const std::exception& e_) { g_err = std::string("hlala_typer_write_summary: ") + e_.what(); return HLALA_E_ARG; }

static int write_pairs_table(const hlala_locus* L, const int C, const int32_t* order, const double* p_normalized, const double* pair_ll, const double* mis_avg, const std::string& dir)
{
    const std::string& locus = L->name;
    // ---- all pairs, :2451-2488
    const long long nPairs = (long long)C * (C + 1) / 2;
    std::vector<int> c1Of((size_t)nPairs), c2Of((size_t)nPairs);
    { long long i = 0; for(int a = 0; a < C; a++) for(int b = a; b < C; b++, i++) { c1Of[(size_t)i] = a; c2Of[(size_t)i] = b; } }
    {
        std::ofstream ap((dir + "/R1_PP_" + locus + "_pairs.txt").c_str());
        if(!ap.is_open()) return fail(HLALA_E_ARG, "cannot write " + dir + "/R1_PP_" + locus + "_pairs.txt");
        ap << "ClusterID" << "\t" << "P" << "\t" << "LL" << "\t" << "Mismatches_avg" << "\n";
        for(long long k = 0; k < nPairs; k++) if(order[k] < 0 || order[k] >= nPairs) return fail(HLALA_E_ARG, "hlala_locus_write_pairs_file: order[] holds an index outside the pair table");
        // one line per cluster pair (millions for a class-I locus): the lines are formatted by the same iostream calls as the reference's, chunk by chunk on all
        // host threads, and written in order
        const long long CH = 65536; const long long nChunks = (nPairs + CH - 1) / CH;
        std::vector<std::string> parts((size_t)nChunks);
        std::atomic<long long> next(0);
        auto work = [&]() {
            for(;;) {
                const long long c = next.fetch_add(1); if(c >= nChunks) break;
                // (a double goes out as the stream's default format -- %g, six significant digits -- through std::to_chars(general, 6), which is defined
                // as that printf conversion and six times as fast as operator<<; a value that is not finite takes the stream, whose spelling of it is the reference's)
                std::string& out = parts[(size_t)c];
                const long long k1 = std::min(nPairs, (c + 1) * CH);
                out.reserve((size_t)(k1 - c * CH) * 64);
                auto num = [&](double v) {
                    if(std::isfinite(v)) { char b[40]; const auto r = std::to_chars(b, b + sizeof b, v, std::chars_format::general, 6); out.append(b, r.ptr); }
                    else { std::ostringstream os; os << v; out += os.str(); }
                };
                for(long long k = c * CH; k < k1; k++) {
                    const int cI = order[k];
                    // (the lines come in `order`, their values lie at the pair's own index: five cache misses per line, asked for sixteen lines ahead)
                    if(k + 16 < k1) { const int cF = order[k + 16]; __builtin_prefetch(&c1Of[cF]); __builtin_prefetch(&c2Of[cF]); __builtin_prefetch(&p_normalized[cF]); __builtin_prefetch(&pair_ll[cF]); __builtin_prefetch(&mis_avg[cF]); }
                    out += L->clusterId[c1Of[cI]]; out += '/'; out += L->clusterId[c2Of[cI]]; out += '\t';
                    num(p_normalized[cI]); out += '\t'; num(pair_ll[cI]); out += '\t'; num(mis_avg[cI]); out += '\n';
                }
            }
        };
        unsigned T = (unsigned)hlala_host::host_cpu_budget(); if(T > 64) T = 64; if((long long)T > nChunks) T = (unsigned)nChunks;
        std::exception_ptr werr; std::mutex wm;
        auto guarded = [&]() { try { work(); } catch(...) { std::lock_guard<std::mutex> g(wm); if(!werr) werr = std::current_exception(); next.store(nChunks); } };
        std::vector<std::thread> th;
        try { for(unsigned t = 1; t < T; t++) th.emplace_back(guarded); } catch(...) { next.store(nChunks); for(std::thread& x : th) x.join(); throw; }
        guarded();
        for(std::thread& x : th) x.join();
        if(werr) std::rethrow_exception(werr);
        for(const std::string& pstr : parts) ap.write(pstr.data(), (std::streamsize)pstr.size());
    }
    return HLALA_OK;
}

extern "C" int hlala_locus_write_pairs_file(const hlala_locus* L, int32_t n_clusters, const int32_t* order, const double* p_normalized, const double* pair_ll,
                                            const double* mis_avg, const char* out_dir)
{
    if(!L || !order || !p_normalized || !pair_ll || !mis_avg || !out_dir) return fail(HLALA_E_ARG, "hlala_locus_write_pairs_file: null argument");
    if(n_clusters != (int32_t)L->clusterId.size()) return fail(HLALA_E_ARG, "hlala_locus_write_pairs_file: n_clusters does not match the locus");
    try { return write_pairs_table(L, n_clusters, order, p_normalized, pair_ll, mis_avg, out_dir); }
    catch(const std::exception& e) { return fail(HLALA_E_ARG, std::string("hlala_locus_write_pairs_file: ") + e.what()); }
}

extern "C" int hlala_locus_write_files(const hlala_locus* L, const hlala_locus_report_in* in, const char* out_dir, hlala_locus_report_out* res)
try {
    if(!L || !in || !out_dir || !in->pos || !in->filter || !in->call) return fail(HLALA_E_ARG, "hlala_locus_write_files: null argument");
    const hlala_exon_positions_out* pos = in->pos;
    const int nReads = pos->n_reads, nPos = pos->n_pos, P = L->P, C = L->C;
    if(in->n_clusters != C) return fail(HLALA_E_ARG, "hlala_locus_write_files: n_clusters does not match the locus");
    if(nReads > 0 && (!pos->read_reverse || !pos->read_mapq || !in->unit_name_1)) return fail(HLALA_E_ARG, "hlala_locus_write_files: read_reverse, read_mapq and unit_name_1 are needed");
    if(!in->pair_ll || !in->mis_avg || !in->mis_min || !in->order || !in->p_normalized) return fail(HLALA_E_ARG, "hlala_locus_write_files: the all-pairs tables are needed");
    // ---- the filters once more, with the per-allele counts of the stage the reports print
    std::vector<uint8_t> use((size_t)std::max(nPos, 1), 0), ignored((size_t)std::max(nReads, 1), 0);
    std::vector<std::vector<hlala_host::AlleleTally>> tallies;
    hlala_filter_stats fs;
    int rc = hlala_host::filter_positions_impl(pos, in->filter, use.data(), ignored.data(), &fs, &tallies);
    if(rc) return fail(rc, "hlala_locus_write_files: filter_positions failed");
    // ---- pile-up buckets per exon column, entries in (read, position) order, :1883-1931
    std::vector<int> readOf((size_t)nPos, 0);
    for(int r = 0; r < nReads; r++) for(int j = pos->pos_off[r]; j < pos->pos_off[r + 1]; j++) readOf[j] = r;
    std::vector<int> bOff((size_t)P + 1, 0);
    auto piled = [&](int j) { return use[j] && !(in->long_read_mode && pos->pos_novel_gap[j] >= 2); };
    for(int j = 0; j < nPos; j++) { if(pos->pos_exon[j] < 0 || pos->pos_exon[j] >= P) return fail(HLALA_E_ARG, "hlala_locus_write_files: exon position outside the locus"); if(piled(j)) bOff[(size_t)pos->pos_exon[j] + 1]++; }
    for(int c = 0; c < P; c++) bOff[(size_t)c + 1] += bOff[c];
    std::vector<int> pile((size_t)bOff[P]);
    { std::vector<int> fill(bOff.begin(), bOff.end() - 1); for(int j = 0; j < nPos; j++) if(piled(j)) pile[(size_t)fill[pos->pos_exon[j]]++] = j; }
    auto genotype = [&](int j) { return std::string((const char*)pos->geno_chars + pos->geno_off[j], (size_t)(pos->geno_off[j + 1] - pos->geno_off[j])); };
    auto tally = [&](int col, const std::string& a) -> const hlala_host::AlleleTally* {
        if(col >= (int)tallies.size()) return nullptr;
        for(const auto& t : tallies[col]) if(t.count > 0 && t.allele == a) return &t;
        return nullptr;
    };
    const std::string dir(out_dir), locus = L->name;
    std::set<std::string> utilized;
    if(in->unit_stats) {
        // histogram_matchesPerRead.txt: per pair that passes the pair test of the locus (:1404-1429), then per piled position (:1928)
        const hlala_unit_stats_out* us = in->unit_stats;
        std::ofstream hst((dir + "/histogram_matchesPerRead.txt").c_str(), std::ios::app);
        if(!hst.is_open()) return fail(HLALA_E_ARG, "cannot append to " + dir + "/histogram_matchesPerRead.txt (hlala_typer_begin_output first)");
        if(!in->long_read_mode && in->unit_name_2)
            for(int32_t u = 0; u < in->n_units; u++) {
                if((in->unit_mask && !in->unit_mask[u]) || !us->valid[u]) continue;
                const double w1 = us->weighted_ok[2 * u], w2 = us->weighted_ok[2 * u + 1];
                if(us->strands_valid[u] && (std::abs((double)us->distance[u] - in->insert_mean) <= (5 * in->insert_sd)) && (us->mate_mapq[2 * u] >= in->min_mapq) &&
                   ((w1 >= in->min_weighted_ok) && (w2 >= in->min_weighted_ok))) {
                    hst << locus << "\t" << "read" << w1 << "\n";
                    hst << locus << "\t" << "read" << w2 << "\n";
                    hst << locus << "\t" << "readPair" << (w1 + w2) / 2.0 << "\n";
                }
            }
        for(int j = 0; j < nPos; j++) if(piled(j)) hst << locus << "\t" << "base" << pos->read_weighted_ok[2 * readOf[j] + (pos->pos_mate[j] == 2 ? 1 : 0)] << "\n";
    }
    {
        std::ofstream pu((dir + "/R1_pileup_" + locus + ".txt").c_str());
        if(!pu.is_open()) return fail(HLALA_E_ARG, "cannot write " + dir + "/R1_pileup_" + locus + ".txt");
        std::vector<int> exonFirstCol(L->exonLength.size(), 0);
        for(size_t e = 1; e < L->exonLength.size(); e++) exonFirstCol[e] = exonFirstCol[e - 1] + L->exonLength[e - 1];
        // the columns that get a line, in the order of the file; their lines are formatted side by side (a class-I locus at 30x piles half a million positions,
        // a dozen numbers each) in runs of columns of about equal numbers of piled positions, written one run after the other
        struct ColRef { int e, col, c0; };
        std::vector<ColRef> cols;
        for(size_t e = 0; e < L->exonLength.size(); e++) {
            const int c0 = exonFirstCol[e], c1 = c0 + L->exonLength[e];
            if(bOff[c1] == bOff[c0]) continue;                               // an exon without any piled position is not in pileUpPerPosition: no lines
            for(int col = c0; col < c1; col++) cols.push_back(ColRef{(int)e, col, c0});
        }
        long long piledTotal = 0; for(const ColRef& cr : cols) piledTotal += bOff[(size_t)cr.col + 1] - bOff[cr.col];
        long long perRun = 20000; if(const char* ev = getenv("HLALA_PILEUP_RUN")) { const long long v = atoll(ev); if(v > 0) perRun = v; }       // (tests: small runs)
        const int K = (int)std::max<long long>(1, std::min<long long>(std::min<long long>(8, std::max(2, hlala_host::host_cpu_budget())), piledTotal / perRun));
        std::vector<size_t> cut((size_t)K + 1, cols.size()); cut[0] = 0;
        { long long run = 0; int k = 1; for(size_t i = 0; i < cols.size() && k < K; i++) { run += bOff[(size_t)cols[i].col + 1] - bOff[cols[i].col]; if(run >= piledTotal * k / K) cut[(size_t)k++] = i + 1; } }
        std::vector<std::string> text((size_t)K); std::vector<std::set<std::string>> used((size_t)K); std::vector<int> bad((size_t)K, 0);
        auto format_run = [&](int k) {
            std::string& out = text[(size_t)k];
            for(size_t ci = cut[(size_t)k]; ci < cut[(size_t)k + 1]; ci++) {
                const int e = cols[ci].e, col = cols[ci].col, c0 = cols[ci].c0;
                const int n = bOff[(size_t)col + 1] - bOff[col];
                out += to_str(e); out += '\t'; out += to_str(col - c0); out += '\t'; out += to_str(n);
                if(n == 0) { out += '\n'; continue; }
                std::map<std::string, std::vector<int>> alleleCounts;
                std::string all;
                for(int i = 0; i < n; i++) {
                    const int j = pile[(size_t)bOff[col] + i], r = readOf[j], m = pos->pos_mate[j] == 2 ? 1 : 0;
                    const std::string g = genotype(j);
                    std::string q;
                    for(int k2 = pos->geno_off[j]; k2 < pos->geno_off[j + 1]; k2++) {
                        if(pos->geno_chars[k2] == '_') continue;                                          // a gap carries no quality
                        if(!q.empty()) q += ", ";
                        q += to_str((int)(char)pos->qual_chars[k2]);
                    }
                    const int unit = pos->read_pair[r];
                    const char* n1 = in->unit_name_1[unit]; const char* n2 = in->unit_name_2 ? in->unit_name_2[unit] : "";
                    const std::string thisID = m ? n2 : n1, otherID = in->unit_name_2 ? (m ? n1 : n2) : "";
                    if(i) all += ", ";
                    all += g + " (" + q + ")" + " [" + "pairsDistance " + to_str((double)pos->read_distance[r]) + " | " + "alignmentLength " + to_str(pos->read_cols_nongap[2 * r + m]) + " | " +
                           to_str(phred_to_pcorrect(pos->pos_mapq[j])) + " | " + to_str(pos->read_mapq[2 * r + m]) + " " + to_str(pos->read_mapq[2 * r + m]) + " | " +
                           to_str(pos->read_weighted_ok[2 * r + m]) + " " + to_str(pos->read_weighted_ok[2 * r + (1 - m)]) + " | " + thisID + " " + otherID + "]";
                    used[(size_t)k].insert(thisID);
                    alleleCounts[g].push_back(pos->read_cols_nongap[2 * r + m]);
                }
                std::string summary;
                for(const auto& a : alleleCounts) {
                    long long sum = 0; for(int l : a.second) sum += l;
                    const double avgL = (double)sum / (double)a.second.size();
                    const hlala_host::AlleleTally* t = tally(col, a.first);
                    if(!t) { bad[(size_t)k] = 1; return; }
                    const int minStrand = std::min(t->reverse, t->count - t->reverse);
                    summary += a.first + "x" + to_str(a.second.size()) + "[" + to_str(avgL) + ";" + to_str((double)minStrand / (double)t->count) + ";" + to_str((double)t->from_first / (double)t->count) + "]";
                }
                out += '\t'; out += all; out += '\t'; out += summary; out += '\n';
            }
        };
        if(K == 1) format_run(0);
        else {
            std::vector<std::thread> th; std::vector<std::exception_ptr> ex((size_t)K);
            try { for(int k = 0; k < K; k++) th.emplace_back([&, k]() { try { format_run(k); } catch(...) { ex[(size_t)k] = std::current_exception(); } }); }
            catch(...) { for(std::thread& t : th) t.join(); throw; }
            for(std::thread& t : th) t.join();
            for(const std::exception_ptr& e : ex) if(e) std::rethrow_exception(e);
        }
        for(int k = 0; k < K; k++) {
            if(bad[(size_t)k]) return fail(HLALA_E_STATE, "hlala_locus_write_files: piled allele without counts");
            pu.write(text[(size_t)k].data(), (std::streamsize)text[(size_t)k].size());
            utilized.insert(used[(size_t)k].begin(), used[(size_t)k].end());
        }
        std::ofstream ids((dir + "/R1_readIDs_" + locus + ".txt").c_str());
        if(!ids.is_open()) return fail(HLALA_E_ARG, "cannot write " + dir + "/R1_readIDs_" + locus + ".txt");
        for(const std::string& id : utilized) ids << id << "\n";
    }
    // ---- all pairs, :2451-2488
    if(!in->pairs_file_done) { const int rcp = write_pairs_table(L, C, in->order, in->p_normalized, in->pair_ll, in->mis_avg, dir); if(rcp != HLALA_OK) return rcp; }
    // ---- coverage, column incompatibilities, best guesses, :2543-2759
    const int first = in->call->first_cluster, second = in->call->second_cluster;
    if(first < 0 || first >= C || second < 0 || second >= C) return fail(HLALA_E_ARG, "hlala_locus_write_files: called clusters outside the locus");
    const uint8_t* s1 = L->clusterSeq.data() + (size_t)first * P; const uint8_t* s2 = L->clusterSeq.data() + (size_t)second * P;
    std::vector<double> positionalCoverages((size_t)P);
    for(int col = 0; col < P; col++) positionalCoverages[col] = bOff[(size_t)col + 1] - bOff[col];
    std::sort(positionalCoverages.begin(), positionalCoverages.end(), std::less<int>());                // sic: the reference's comparator converts to int
    size_t allTotal = 0, allIncompatible = 0; int unaccounted = 0;
    std::vector<int> colTotal((size_t)P), colIncompatible((size_t)P);
    for(int col = 0; col < P; col++) {
        const std::string a1(1, (char)s1[col]), a2(1, (char)s2[col]);
        int total = 0, bad = 0;
        for(int i = bOff[col]; i < bOff[(size_t)col + 1]; i++) { total++; const std::string g = genotype(pile[i]); if(g != a1 && g != a2) bad++; }
        allTotal += total; allIncompatible += bad; colTotal[col] = total; colIncompatible[col] = bad;
        if(col < (int)tallies.size()) {
            int totalCoverage = 0; bool any = false;
            for(const auto& t : tallies[col]) if(t.count > 0 && t.post_filtering >= 0) { totalCoverage += t.post_filtering; any = true; }
            if(any && totalCoverage >= in->unaccounted_min_coverage)
                for(const auto& t : tallies[col]) if(t.count > 0 && t.post_filtering >= 0) {
                    if(t.allele == a1 || t.allele == a2) continue;
                    if((double)t.post_filtering / (double)totalCoverage >= in->unaccounted_min_fraction) unaccounted++;
                }
        }
    }
    const double avgErr = allTotal > 0 ? (double)allIncompatible / (double)allTotal : 0;
    double minP = -1;
    {
        std::ofstream ce((dir + "/R1_columnIncompatibilities_" + locus + ".txt").c_str());
        if(!ce.is_open()) return fail(HLALA_E_ARG, "cannot write " + dir + "/R1_columnIncompatibilities_" + locus + ".txt");
        ce << "Column" << "\t" << "Coverage" << "\t" << "ExpectedIncompatible" << "\t" << "ObservedIncompatible" << "\t" << "p" << "\n";
        for(int col = 0; col < P; col++) {
            const int cov = colTotal[col], obs = colIncompatible[col];
            const double expected = avgErr * cov;
            double p = 1;
            if(obs > expected) { const double o[2] = {(double)(cov - obs), (double)obs}, e[2] = {cov - expected, expected}; p = chi_sq_p(o, e); }
            ce << to_str(col) << "\t" << to_str(cov) << "\t" << to_str(expected) << "\t" << to_str(obs) << "\t" << to_str(p) << "\n";
            if(minP < 0 || p < minP) minP = p;
        }
    }
    const double locusCoverage = (double)fs.bases_used / (double)P;
    const double firstDecile = positionalCoverages[(size_t)(int)((double)positionalCoverages.size() / 10.0)], minimumCoverage = positionalCoverages[0];
    const double q1First = in->call->first_marginal, q1Second = in->call->second_p;
    const long long fsIdx = (long long)std::min(first, second) * C - (long long)std::min(first, second) * (std::min(first, second) - 1) / 2 + (std::max(first, second) - std::min(first, second));
    const double q2 = -1 * in->mis_min[fsIdx];                                                          // bestGuess_secondAllele.first, :2527-2533
    {
        std::ofstream bg((dir + "/R1_bestguess.txt").c_str(), std::ios::app);
        if(!bg.is_open()) return fail(HLALA_E_ARG, "cannot append to " + dir + "/R1_bestguess.txt (hlala_typer_begin_output first)");
        bg << locus << "\t" << 1 << "\t" << L->clusterId[first] << "\t" << q1First << "\t" << q2 << "\t" << locusCoverage << "\t" << firstDecile << "\t" << minimumCoverage << "\t" << in->kmers_covered[0] << "\t" << avgErr << "\t" << unaccounted << "\n";
        bg << locus << "\t" << 2 << "\t" << L->clusterId[second] << "\t" << q1Second << "\t" << q2 << "\t" << locusCoverage << "\t" << firstDecile << "\t" << minimumCoverage << "\t" << in->kmers_covered[1] << "\t" << avgErr << "\t" << unaccounted << "\n";
    }
    if(L->typer && L->typer->gLoci.count(locus)) {                                                      // can_translateToG_locus
        bool p1 = false, p2 = false;
        const std::string g1 = to_g_group(*L->typer, L->members[first], p1), g2 = to_g_group(*L->typer, L->members[second], p2);
        std::ofstream bg((dir + "/R1_bestguess_G.txt").c_str(), std::ios::app);
        if(!bg.is_open()) return fail(HLALA_E_ARG, "cannot append to " + dir + "/R1_bestguess_G.txt (hlala_typer_begin_output first)");
        bg << locus << "\t" << 1 << "\t" << g1 << "\t" << q1First << "\t" << q2 << "\t" << locusCoverage << "\t" << firstDecile << "\t" << minimumCoverage << "\t" << in->kmers_covered[0] << "\t" << avgErr << "\t" << unaccounted << "\t" << p1 << "\n";
        bg << locus << "\t" << 2 << "\t" << g2 << "\t" << q1Second << "\t" << q2 << "\t" << locusCoverage << "\t" << firstDecile << "\t" << minimumCoverage << "\t" << in->kmers_covered[1] << "\t" << avgErr << "\t" << unaccounted << "\t" << p2 << "\n";
    }
    if(res) {
        res->locus_coverage = locusCoverage; res->first_decile_coverage = firstDecile; res->minimum_coverage = minimumCoverage; res->avg_column_error = avgErr;
        res->min_column_p = minP; res->n_columns_unaccounted = unaccounted; res->n_utilized_reads = (int32_t)utilized.size(); res->bases_used = fs.bases_used;
        res->n_piled_positions = bOff[P];
    }
    return HLALA_OK;
} catch(const std::exception& e_) { g_err = std::string("hlala_locus_write_files: ") + e_.what(); return HLALA_E_ARG; }
