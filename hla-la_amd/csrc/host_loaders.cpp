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
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
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
{
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
}

extern "C" int hlala_graph_file_desc(const hlala_graph_file* g, hlala_graph_desc* d)
{
    if(!g || !d) return HLALA_E_ARG;
    d->n_levels = g->n_levels; d->n_nodes = (int32_t)g->node_level.size(); d->n_edges = (int32_t)g->edge_from.size();
    d->node_level = g->node_level.data(); d->edge_from = g->edge_from.data(); d->edge_to = g->edge_to.data(); d->edge_label = g->edge_label.data();
    return HLALA_OK;
}

extern "C" void hlala_graph_file_free(hlala_graph_file* g) { delete g; }

// ---- binary cache: header {magic, n_levels, n_nodes, n_edges}, then node_level, edge_from, edge_to (int32) and edge_label (bytes)
static const char CACHE_MAGIC[8] = {'H', 'L', 'A', 'L', 'A', 'G', 'R', '1'};

extern "C" int hlala_graph_cache_save(const hlala_graph_desc* d, const char* path)
{
    if(!d || !path) return HLALA_E_ARG;
    FILE* f = fopen(path, "wb");
    if(!f) return fail(std::string("Cannot open cache file for writing: ") + path);
    int32_t hdr[3] = {d->n_levels, d->n_nodes, d->n_edges};
    bool ok = fwrite(CACHE_MAGIC, 1, 8, f) == 8 && fwrite(hdr, 4, 3, f) == 3 &&
              fwrite(d->node_level, 4, (size_t)d->n_nodes, f) == (size_t)d->n_nodes && fwrite(d->edge_from, 4, (size_t)d->n_edges, f) == (size_t)d->n_edges &&
              fwrite(d->edge_to, 4, (size_t)d->n_edges, f) == (size_t)d->n_edges && fwrite(d->edge_label, 1, (size_t)d->n_edges, f) == (size_t)d->n_edges;
    ok = (fclose(f) == 0) && ok;
    return ok ? HLALA_OK : fail(std::string("short write to ") + path);
}

extern "C" int hlala_graph_cache_load(const char* path, hlala_graph_file** out)
{
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
}
