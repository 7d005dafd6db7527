// flat_graph.cpp -- see flat_graph.hpp.  Host-only, runs once per graph.
#include "flat_graph.hpp"

#include <algorithm>
#include <map>
#include <numeric>

namespace hlala {

namespace {
struct TrieEnt { int32_t parent; int32_t edge; int32_t len; };
}

std::string flatten_graph(const hlala_graph_desc* g, const hlala_contigs_desc* c, FlatGraph& F, bool keep_unit_jumps)
{
    if(!g || g->n_levels < 2 || g->n_nodes < 2 || g->n_edges < 1) return "graph needs >= 2 levels, nodes and >= 1 edge";
    const int32_t L = g->n_levels, N = g->n_nodes, E = g->n_edges;
    F.L = L; F.N = N; F.E = E;

    // ---- nodes: level-major renumbering, stable in creation order (rank z of alignerBase.cpp:27-37)
    F.level_off.assign(L + 1, 0);
    for(int32_t n = 0; n < N; n++) {
        int32_t l = g->node_level[n];
        if(l < 0 || l >= L) return "node level out of range";
        F.level_off[l + 1]++;
    }
    for(int32_t l = 0; l < L; l++) {
        if(F.level_off[l + 1] == 0) return "level without nodes";
        F.max_nodes_per_level = std::max(F.max_nodes_per_level, F.level_off[l + 1]);
        F.level_off[l + 1] += F.level_off[l];
    }
    F.node_orig.assign(N, 0); F.node_new.assign(N, 0); F.node_level.assign(N, 0);
    {
        std::vector<int32_t> cursor(F.level_off.begin(), F.level_off.end() - 1);
        for(int32_t n = 0; n < N; n++) {
            int32_t id = cursor[g->node_level[n]]++;
            F.node_orig[id] = n; F.node_new[n] = id; F.node_level[id] = g->node_level[n];
        }
    }

    // ---- edges: CSR in edge creation order (std::set<Edge*> order of Node::Outgoing_/Incoming_Edges)
    F.out_off.assign(N + 1, 0); F.in_off.assign(N + 1, 0);
    F.edge_from_new.assign(E, 0); F.edge_to_new.assign(E, 0);
    for(int32_t e = 0; e < E; e++) {
        int32_t a = g->edge_from[e], b = g->edge_to[e];
        if(a < 0 || a >= N || b < 0 || b >= N) return "edge endpoint out of range";
        if(g->node_level[b] != g->node_level[a] + 1) return "edge does not connect consecutive levels";
        if(g->edge_label[e] == 0) return "edge with emission 0";
        F.edge_from_new[e] = F.node_new[a]; F.edge_to_new[e] = F.node_new[b];
        F.out_off[F.node_new[a] + 1]++; F.in_off[F.node_new[b] + 1]++;
    }
    for(int32_t n = 0; n < N; n++) {
        F.max_out_degree = std::max(F.max_out_degree, F.out_off[n + 1]);
        F.max_in_degree = std::max(F.max_in_degree, F.in_off[n + 1]);
        // the reference asserts every node it touches has neighbours (alignerBase.cpp:161, 185)
        if(F.node_level[n] < L - 1 && F.out_off[n + 1] == 0) return "non-terminal node without outgoing edge";
        if(F.node_level[n] > 0 && F.in_off[n + 1] == 0) return "non-initial node without incoming edge";
        F.out_off[n + 1] += F.out_off[n]; F.in_off[n + 1] += F.in_off[n];
    }
    F.out_to.assign(E, 0); F.out_label.assign(E, 0); F.out_eid.assign(E, 0);
    F.in_from.assign(E, 0); F.in_label.assign(E, 0); F.in_eid.assign(E, 0);
    {
        std::vector<int32_t> co(F.out_off.begin(), F.out_off.end() - 1), ci(F.in_off.begin(), F.in_off.end() - 1);
        for(int32_t e = 0; e < E; e++) {
            int32_t a = F.edge_from_new[e], b = F.edge_to_new[e];
            int32_t io = co[a]++, ii = ci[b]++;
            F.out_to[io] = b; F.out_label[io] = g->edge_label[e]; F.out_eid[io] = e;
            F.in_from[ii] = a; F.in_label[ii] = g->edge_label[e]; F.in_eid[ii] = e;
        }
    }

    // ---- gap stretches (processBAM.cpp:91-149): runs of >= 3 levels that have a '_' edge
    std::vector<uint8_t> levelHasGap(L, 0);
    for(int32_t e = 0; e < E; e++) if(g->edge_label[e] == '_') levelHasGap[g->node_level[g->edge_from[e]]] = 1;
    F.gap_stretch.assign(L - 1, 0);
    for(int32_t l = 0; l < L - 1;) {
        if(!levelHasGap[l]) { l++; continue; }
        int32_t s = l;
        while(l < L - 1 && levelHasGap[l]) l++;
        if(l - s >= 3) std::fill(F.gap_stretch.begin() + s, F.gap_stretch.begin() + l, 1);
    }

    // ---- gap-edge paths (Graph.cpp:347-476).  Running paths live in a trie (parent entry + edge)
    // instead of the reference's per-level vector copies; iteration orders are the reference's
    // std::map<Node*,...> orders with creation index for the pointer.
    std::vector<TrieEnt> trie;
    std::vector<int32_t> completed;                       // trie entries, in completedGapEdgePaths order
    std::map<int32_t, std::map<int32_t, int32_t>> running, next;   // node(orig) -> from(orig) -> trie entry
    std::vector<int32_t> levelEdges;
    for(int32_t l = 0; l < L; l++) {
        if(running.empty() && !levelHasGap[l]) continue;
        next.clear();
        for(auto& nodeIt : running) {
            int32_t n = F.node_new[nodeIt.first];
            int nonGap = 0;
            for(int32_t i = F.out_off[n]; i < F.out_off[n + 1]; i++) {
                if(F.out_label[i] == '_') {
                    int32_t t = F.node_orig[F.out_to[i]];
                    auto& slot = next[t];
                    for(auto& fromIt : nodeIt.second)
                        if(slot.find(fromIt.first) == slot.end()) {
                            trie.push_back({fromIt.second, F.out_eid[i], trie[fromIt.second].len + 1});
                            slot[fromIt.first] = (int32_t)trie.size() - 1;
                        }
                } else nonGap++;
            }
            if(nonGap != 0 || l == L - 1)
                for(auto& fromIt : nodeIt.second) completed.push_back(fromIt.second);
        }
        if(levelHasGap[l]) {
            levelEdges.clear();
            for(int32_t n = F.level_off[l]; n < F.level_off[l + 1]; n++)
                for(int32_t i = F.out_off[n]; i < F.out_off[n + 1]; i++)
                    if(F.out_label[i] == '_') levelEdges.push_back(F.out_eid[i]);
            std::sort(levelEdges.begin(), levelEdges.end());
            for(int32_t e : levelEdges) {
                int32_t from = g->edge_from[e], to = g->edge_to[e];
                if(running.find(from) != running.end()) continue;          // "seen_gap_edge"
                auto& slot = next[to];
                if(slot.find(from) == slot.end()) {
                    trie.push_back({-1, e, 1});
                    slot[from] = (int32_t)trie.size() - 1;
                }
            }
        }
        running.swap(next);
    }
    const int32_t P = (int32_t)completed.size();
    F.path_first.assign(P, 0); F.path_last.assign(P, 0); F.path_len.assign(P, 0); F.path_off.assign(P + 1, 0);
    for(int32_t p = 0; p < P; p++) F.path_off[p + 1] = F.path_off[p] + trie[completed[p]].len;
    F.path_edges.assign(F.path_off[P], 0);
    for(int32_t p = 0; p < P; p++) {
        int32_t ent = completed[p], len = trie[ent].len;
        for(int32_t k = len - 1; k >= 0; k--) { F.path_edges[F.path_off[p] + k] = trie[ent].edge; ent = trie[ent].parent; }
        F.path_len[p] = len;
        F.path_first[p] = F.edge_from_new[F.path_edges[F.path_off[p]]];
        F.path_last[p] = F.edge_to_new[F.path_edges[F.path_off[p] + len - 1]];
    }
    // jump tables (gapEdgePaths_connectedNodes_{forwards,backwards}): per node, ordered by the
    // creation index of the node at the other end (std::map<Node*,Edge*> order)
    {
        std::vector<int32_t> idx(P);
        std::iota(idx.begin(), idx.end(), 0);
        auto build = [&](const std::vector<int32_t>& key, const std::vector<int32_t>& other, std::vector<int32_t>& off,
                         std::vector<int32_t>& node, std::vector<int32_t>& path) -> bool {
            std::sort(idx.begin(), idx.end(), [&](int32_t a, int32_t b) {
                if(key[a] != key[b]) return key[a] < key[b];
                return F.node_orig[other[a]] < F.node_orig[other[b]]; });
            off.assign(N + 1, 0); node.assign(P, 0); path.assign(P, 0);
            for(int32_t i = 0; i < P; i++) {
                int32_t p = idx[i];
                if(i > 0 && key[idx[i - 1]] == key[p] && other[idx[i - 1]] == other[p]) return false;   // Graph.cpp:461-462 asserts
                off[key[p] + 1]++; node[i] = other[p]; path[i] = p;
            }
            for(int32_t n = 0; n < N; n++) { F.max_jumps = std::max(F.max_jumps, off[n + 1]); off[n + 1] += off[n]; }
            return true;
        };
        if(!build(F.path_first, F.path_last, F.jf_off, F.jf_node, F.jf_path)) return "duplicate gap-edge path between two nodes";
        if(!build(F.path_last, F.path_first, F.jb_off, F.jb_node, F.jb_path)) return "duplicate gap-edge path between two nodes";
        F.jf_lvl.resize(P); F.jb_lvl.resize(P);
        for(int32_t i = 0; i < P; i++) { F.jf_lvl[i] = F.node_level[F.jf_node[i]]; F.jb_lvl[i] = F.node_level[F.jb_node[i]]; }
        // the tables the device walks: without the one-edge paths (flat_graph.hpp)
        auto filt = [&](const std::vector<int32_t>& off, const std::vector<int32_t>& node, const std::vector<int32_t>& path, const std::vector<int32_t>& lvl,
                        std::vector<int32_t>& doff, std::vector<int32_t>& dnode, std::vector<int32_t>& dpath, std::vector<int32_t>& dlvl) {
            doff.assign(N + 1, 0); dnode.clear(); dpath.clear(); dlvl.clear();
            for(int32_t n = 0; n < N; n++) {
                for(int32_t i = off[n]; i < off[n + 1]; i++)
                    if(keep_unit_jumps || F.path_len[path[i]] >= 2) { dnode.push_back(node[i]); dpath.push_back(path[i]); dlvl.push_back(lvl[i]); }
                doff[n + 1] = (int32_t)dnode.size();
            }
        };
        filt(F.jf_off, F.jf_node, F.jf_path, F.jf_lvl, F.djf_off, F.djf_node, F.djf_path, F.djf_lvl);
        filt(F.jb_off, F.jb_node, F.jb_path, F.jb_lvl, F.djb_off, F.djb_node, F.djb_path, F.djb_lvl);
    }

    // ---- level -> (sequence id, position), last writer wins (processBAM.cpp:4455)
    F.lp_off.assign(L + 1, 0);
    if(c && c->n_contigs > 0) {
        const int64_t total = c->contig_off[c->n_contigs];
        // contigs visited in ascending sequence id so each level's entries come out sorted by id
        std::vector<int32_t> order(c->n_contigs);
        std::iota(order.begin(), order.end(), 0);
        std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) { return c->contig_seqid[a] < c->contig_seqid[b]; });
        for(int64_t p = 0; p < total; p++) {
            int32_t l = c->contig_level[p];
            if(l < 0 || l >= L) return "translation level out of range";
        }
        // pass 1: count distinct (level, contig) -- positions of one contig mapping to the same level collapse
        std::vector<int32_t> lastContigAtLevel(L, -1);
        for(int32_t oc : order)
            for(int64_t p = c->contig_off[oc]; p < c->contig_off[oc + 1]; p++) {
                int32_t l = c->contig_level[p];
                if(lastContigAtLevel[l] != oc) { lastContigAtLevel[l] = oc; F.lp_off[l + 1]++; }
            }
        for(int32_t l = 0; l < L; l++) F.lp_off[l + 1] += F.lp_off[l];
        F.lp_seqid.assign(F.lp_off[L], 0); F.lp_pos.assign(F.lp_off[L], 0);
        std::vector<int64_t> cursor(F.lp_off.begin(), F.lp_off.end() - 1);
        std::fill(lastContigAtLevel.begin(), lastContigAtLevel.end(), -1);
        for(int32_t oc : order)
            for(int64_t p = c->contig_off[oc]; p < c->contig_off[oc + 1]; p++) {
                int32_t l = c->contig_level[p];
                if(lastContigAtLevel[l] != oc) {
                    lastContigAtLevel[l] = oc;
                    // two contigs with the same sequence id overwrite each other in the reference's map
                    if(cursor[l] > F.lp_off[l] && F.lp_seqid[cursor[l] - 1] == c->contig_seqid[oc]) cursor[l]--;
                    F.lp_seqid[cursor[l]] = c->contig_seqid[oc];
                    F.lp_pos[cursor[l]] = (int32_t)(p - c->contig_off[oc]);
                    cursor[l]++;
                } else {
                    F.lp_pos[cursor[l] - 1] = (int32_t)(p - c->contig_off[oc]);     // later position overwrites
                }
            }
    }
    // ---- ranks among parallel entries (same node, same target), in entry order
    {
        auto ranks = [&](const std::vector<int32_t>& off, const std::vector<int32_t>& to, std::vector<uint8_t>& prank) {
            prank.assign(to.size(), 0);
            std::vector<std::pair<int32_t, int32_t>> tmp;      // (target, entry) of one node
            for(int32_t n = 0; n < N; n++) {
                const int32_t e0 = off[n], deg = off[n + 1] - e0;
                if(deg < 2) continue;
                if(deg <= 8) {                                   // the common case: compare pairwise
                    for(int32_t a = 1; a < deg; a++) { int r = 0; for(int32_t b = 0; b < a; b++) if(to[e0 + b] == to[e0 + a]) r++; if(r > 255) r = 255; prank[e0 + a] = (uint8_t)r; F.max_parallel = std::max(F.max_parallel, r + 1); }
                } else {
                    tmp.clear(); for(int32_t a = 0; a < deg; a++) tmp.emplace_back(to[e0 + a], a);
                    std::sort(tmp.begin(), tmp.end());
                    for(size_t i = 0; i < tmp.size();) { size_t j = i; while(j < tmp.size() && tmp[j].first == tmp[i].first) { const int r = (int)(j - i); prank[e0 + tmp[j].second] = (uint8_t)std::min(r, 255); F.max_parallel = std::max(F.max_parallel, r + 1); j++; } i = j; }
                }
            }
        };
        F.max_parallel = (E > 0) ? 1 : 0;
        ranks(F.out_off, F.out_to, F.out_prank); ranks(F.in_off, F.in_from, F.in_prank);
        ranks(F.djf_off, F.djf_node, F.jf_prank); ranks(F.djb_off, F.djb_node, F.jb_prank);
    }
    // ---- node records: everything one DP iteration needs of a frontier node in one 32-byte read
    {
        std::string recErr;
        auto build = [&](const std::vector<int32_t>& off, const std::vector<int32_t>& to, const std::vector<uint8_t>& lab,
                         const std::vector<int32_t>& joff, const std::vector<int32_t>& jnode, const std::vector<int32_t>& jlvl,
                         std::vector<int32_t>& rec) {
            rec.assign((size_t)8 * N, 0);
            for(int32_t n = 0; n < N; n++) {
                const int32_t e0 = off[n], deg = off[n + 1] - e0;
                const int32_t j0 = joff[n], nj = joff[n + 1] - j0;
                int32_t* r = &rec[(size_t)8 * n];
                if(deg > 0xFFFF || nj > 0x7FFF) { recErr = "node with more than 65535 edges or 32767 gap-path jumps in one direction (node record layout)"; return; }
                r[0] = e0; r[1] = deg | (int32_t)((uint32_t)nj << 16);
                r[2] = deg > 0 ? to[e0] : 0; r[3] = deg > 1 ? to[e0 + 1] : 0;
                r[4] = j0; r[5] = nj > 0 ? jnode[j0] : 0; r[6] = nj > 0 ? jlvl[j0] : 0;
                r[7] = (deg > 0 ? lab[e0] : 0) | ((deg > 1 ? lab[e0 + 1] : 0) << 8) | ((deg > 1 && to[e0 + 1] == to[e0] ? 1 : 0) << 16);
            }
        };
        // ---- in-edge records of the projection's re-threading DP (flat_graph.hpp)
        F.in_rec.assign((size_t)F.E, 0);
        F.level_fast.assign((size_t)F.L, 0);
        for(int32_t lv = 1; lv < F.L; lv++) {
            const int32_t n0 = F.level_off[lv], n1 = F.level_off[lv + 1];
            if(n1 <= n0) continue;
            const int32_t eFirst = F.in_off[n0], eEnd = F.in_off[n1];
            const int32_t pFirst = F.in_off[F.level_off[lv - 1]];                 // first in-edge of the level before (its edge list starts there)
            bool fast = (n1 - n0) <= 512; int maxd = 1;
            for(int32_t t = n0; t < n1; t++) {
                const int32_t e0 = F.in_off[t], e1 = F.in_off[t + 1];
                if(e1 - e0 < 1 || e1 - e0 > 8) fast = false;
                if(e1 - e0 > maxd) maxd = e1 - e0;
                for(int32_t e = e0; e < e1; e++) {
                    const int32_t from = F.in_from[e], flv = F.node_level[from], fz = from - F.level_off[flv];
                    if(flv != lv - 1 || fz > 511) fast = false;
                    const int32_t own = F.in_off[from + 1] - 1 - pFirst;         // (meaningful when the level before is solved in one slice as well: 0 .. 63)
                    uint32_t r = ((uint32_t)fz & 511u) | (((uint32_t)own & 63u) << 9) | ((uint32_t)std::min(e - e0, 7) << 15) | ((uint32_t)F.in_label[e] << 18);
                    if(e == e1 - 1) r |= 1u << 28;
                    F.in_rec[(size_t)e] = r;
                }
            }
            F.level_fast[(size_t)lv] = fast ? (uint8_t)(((eEnd - eFirst) <= 64 ? 1 : 2) | ((maxd - 1) << 2)) : 0;
        }
        build(F.out_off, F.out_to, F.out_label, F.djf_off, F.djf_node, F.djf_lvl, F.nrec_out);
        build(F.in_off, F.in_from, F.in_label, F.djb_off, F.djb_node, F.djb_lvl, F.nrec_in);
        if(!recErr.empty()) return recErr;
        // ---- how far an extension DP can walk from a level without meeting a gap-path jump (flat_graph.hpp): jfree_out[l] = levels l, l + 1, ... whose nodes
        // have no forward jump, jfree_in[l] = levels l, l - 1, ... without a backward one; capped at 255
        F.jfree_out.assign((size_t)F.L, 0); F.jfree_in.assign((size_t)F.L, 0);
        {
            std::vector<uint8_t> hasF((size_t)F.L, 0), hasB((size_t)F.L, 0);
            for(int32_t n = 0; n < N; n++) { if(F.djf_off[n + 1] > F.djf_off[n]) hasF[(size_t)F.node_level[n]] = 1; if(F.djb_off[n + 1] > F.djb_off[n]) hasB[(size_t)F.node_level[n]] = 1; }
            int run = 255;                                   // (beyond the last level there is nothing to meet)
            for(int32_t l = F.L - 1; l >= 0; l--) { run = hasF[(size_t)l] ? 0 : std::min(255, run + 1); F.jfree_out[(size_t)l] = (uint8_t)run; }
            run = 255;
            for(int32_t l = 0; l < F.L; l++) { run = hasB[(size_t)l] ? 0 : std::min(255, run + 1); F.jfree_in[(size_t)l] = (uint8_t)run; }
            // ---- linear steps (flat_graph.hpp): one node on either side, one to four parallel edges between them, real labels, no gap-path jump along the step
            F.lin_label.assign((size_t)F.L, 0); F.lin_eid.assign((size_t)F.L, -1); F.lin_out.assign((size_t)F.L, 0); F.lin_in.assign((size_t)F.L, 0);
            for(int32_t l = 0; l + 1 < F.L; l++) {
                const int32_t n = F.level_off[l];
                if(F.level_off[l + 1] - n != 1 || F.level_off[l + 2] - F.level_off[l + 1] != 1) continue;
                const int32_t e0 = F.out_off[n], K = F.out_off[n + 1] - e0, i0 = F.in_off[n + 1];
                if(K < 1 || K > 4 || F.in_off[n + 2] - i0 != K) continue;
                if(hasF[(size_t)l] || hasB[(size_t)l + 1]) continue;
                uint32_t w = 0; bool ok = true;
                for(int32_t k = 0; k < K && ok; k++) {
                    // the same edges in the same order from either end: a backward call visits them through the in-CSR of the upper node
                    ok = F.out_to[e0 + k] == n + 1 && F.out_label[e0 + k] != '_' && F.out_label[e0 + k] != 0 && F.in_eid[i0 + k] == F.out_eid[e0 + k];
                    w |= (uint32_t)F.out_label[e0 + k] << (8 * k);
                }
                if(!ok) continue;
                F.lin_label[(size_t)l] = w; F.lin_eid[(size_t)l] = F.out_eid[e0];
            }
            run = 0;
            for(int32_t l = F.L - 1; l >= 0; l--) { run = F.lin_label[(size_t)l] ? std::min(255, run + 1) : 0; F.lin_out[(size_t)l] = (uint8_t)run; }
            run = 0;
            for(int32_t l = 0; l < F.L; l++) { run = (l > 0 && F.lin_label[(size_t)l - 1]) ? std::min(255, run + 1) : 0; F.lin_in[(size_t)l] = (uint8_t)run; }
        }
        // ---- track steps and the jumps along them (flat_graph.hpp; the model of the kernel that uses them: tools/band2/band2_model.cpp build_window / step_word)
        {
            const size_t Lz = (size_t)F.L;
            F.trk_w_out.assign(Lz, 0); F.trk_w_in.assign(Lz, 0); F.trk_out.assign(Lz, 0); F.trk_in.assign(Lz, 0);
            F.trk_j_out.assign(Lz, 0); F.trk_j_in.assign(Lz, 0); F.trk_jp_out.assign(Lz, -1); F.trk_jp_in.assign(Lz, -1);
            auto code = [](uint8_t c) -> int { return c == 'A' ? 0 : (c == 'C' ? 1 : (c == 'G' ? 2 : (c == 'T' ? 3 : (c == 'N' ? 4 : (c == '_' ? 5 : -1))))); };
            auto word = [&](int32_t l, int32_t other, const std::vector<int32_t>& off, const std::vector<int32_t>& to, const std::vector<uint8_t>& lab) -> uint64_t {
                if(other < 0 || other >= F.L) return 0;
                const int32_t n0 = F.level_off[l], nn = F.level_off[l + 1] - n0, o0 = F.level_off[other], on = F.level_off[other + 1] - o0;
                if(nn < 1 || nn > 2 || on < 1 || on > 2) return 0;
                uint64_t w = 0;
                for(int32_t zs = 0; zs < nn; zs++) {
                    const int32_t e0 = off[n0 + zs], deg = off[n0 + zs + 1] - e0;
                    if(deg < 1 || deg > 4) return 0;
                    uint32_t pr[2] = {0, 0}; int firstReal[2] = {-1, -1}, firstGap[2] = {-1, -1};
                    for(int32_t k = 0; k < deg; k++) {
                        const int32_t z = to[e0 + k] - o0; const int cd = code(lab[e0 + k]);
                        if(z < 0 || z > 1 || cd < 0) return 0;
                        pr[z] |= 1u;
                        if(cd == 5) { pr[z] |= 4u; if(firstGap[z] < 0) firstGap[z] = k; }
                        else { pr[z] |= 2u | (1u << (4 + cd)); if(firstReal[z] < 0) firstReal[z] = k; }
                    }
                    for(int z = 0; z < 2; z++) if(firstGap[z] >= 0 && (firstReal[z] < 0 || firstGap[z] < firstReal[z])) pr[z] |= 8u;
                    pr[0] |= (uint32_t)deg << 9;
                    w |= (uint64_t)pr[0] << (16 * (2 * zs)); w |= (uint64_t)pr[1] << (16 * (2 * zs + 1));
                }
                return w;
            };
            auto jumps = [&](int32_t l, const std::vector<int32_t>& off, const std::vector<int32_t>& node, const std::vector<int32_t>& path, uint32_t& jw, int32_t& jp) {
                const int32_t n0 = F.level_off[l], nn = F.level_off[l + 1] - n0;
                int nLong = 0, zA = 0, other = -1, p = -1;
                for(int32_t zs = 0; zs < nn; zs++)
                    for(int32_t i = off[n0 + zs]; i < off[n0 + zs + 1]; i++) if(F.path_len[path[i]] >= 2) { nLong++; zA = zs; other = node[i]; p = path[i]; }
                if(nLong == 0) return;
                const int len = F.path_len[p];
                const int32_t zB = other - F.level_off[F.node_level[other]];
                if(nLong > 1 || len < 4 || len > 29 || zA > 1 || zB > 1) { jw = 2u; return; }
                jw = 1u | ((uint32_t)zA << 2) | ((uint32_t)zB << 3) | ((uint32_t)len << 8); jp = p;
            };
            for(int32_t l = 0; l < F.L; l++) {
                F.trk_w_out[(size_t)l] = word(l, l + 1, F.out_off, F.out_to, F.out_label);
                F.trk_w_in[(size_t)l] = word(l, l - 1, F.in_off, F.in_from, F.in_label);
                jumps(l, F.jf_off, F.jf_node, F.jf_path, F.trk_j_out[(size_t)l], F.trk_jp_out[(size_t)l]);
                jumps(l, F.jb_off, F.jb_node, F.jb_path, F.trk_j_in[(size_t)l], F.trk_jp_in[(size_t)l]);
            }
            int run = 0;
            for(int32_t l = F.L - 1; l >= 0; l--) { run = F.trk_w_out[(size_t)l] ? std::min(255, run + 1) : 0; F.trk_out[(size_t)l] = (uint8_t)run; }
            run = 0;
            for(int32_t l = 0; l < F.L; l++) { run = F.trk_w_in[(size_t)l] ? std::min(255, run + 1) : 0; F.trk_in[(size_t)l] = (uint8_t)run; }
        }
    }
    return "";
}

}  // namespace hlala
