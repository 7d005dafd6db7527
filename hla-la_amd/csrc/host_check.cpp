// host_check.cpp -- host-only view of the one-time graph flatten (no device needed), so the CPU test
// suite can check the host logic of libhlala_gpu.so against the oracle.  Built as libhlala_host.so.
#include <cstring>
#include <string>

#include "flat_graph.hpp"

static thread_local std::string g_err;

extern "C" {
const char* hlala_host_last_error() { return g_err.c_str(); }

hlala::FlatGraph* hlala_host_flatten(const hlala_graph_desc* g, const hlala_contigs_desc* c)
{
    hlala::FlatGraph* F = new hlala::FlatGraph();
    g_err = hlala::flatten_graph(g, c, *F);
    if(!g_err.empty()) { delete F; return nullptr; }
    return F;
}
void hlala_host_free(hlala::FlatGraph* F) { delete F; }

int hlala_host_info(const hlala::FlatGraph* F, hlala_graph_info* info)
{
    memset(info, 0, sizeof(*info));
    info->n_levels = F->L; info->n_nodes = F->N; info->n_edges = F->E; info->n_paths = (int)F->path_len.size();
    info->n_jump_entries = (int64_t)F->jf_node.size(); info->n_path_edges = (int64_t)F->path_edges.size();
    info->n_levelpos_entries = (int64_t)F->lp_seqid.size();
    info->max_nodes_per_level = F->max_nodes_per_level; info->max_out_degree = F->max_out_degree; info->max_in_degree = F->max_in_degree; info->max_jumps = F->max_jumps; info->max_parallel = F->max_parallel;
    for(uint8_t b : F->gap_stretch) info->n_gap_stretch_levels += b;
    return 0;
}
int hlala_host_paths(const hlala::FlatGraph* F, int32_t* first_node, int32_t* last_node, int32_t* length)
{
    for(size_t p = 0; p < F->path_len.size(); p++) { first_node[p] = F->node_orig[F->path_first[p]]; last_node[p] = F->node_orig[F->path_last[p]]; length[p] = F->path_len[p]; }
    return 0;
}
int hlala_host_gap_stretch(const hlala::FlatGraph* F, uint8_t* out) { memcpy(out, F->gap_stretch.data(), F->gap_stretch.size()); return 0; }
// forward jump table of the node with creation index `node`: targets (creation idx) and path ids, in table order
int hlala_host_jumps(const hlala::FlatGraph* F, int node, int forward, int cap, int32_t* target, int32_t* path)
{
    int n = F->node_new[node];
    const auto& off = forward ? F->jf_off : F->jb_off; const auto& nd = forward ? F->jf_node : F->jb_node; const auto& pp = forward ? F->jf_path : F->jb_path;
    int k = 0;
    for(int i = off[n]; i < off[n + 1] && k < cap; i++, k++) { target[k] = F->node_orig[nd[i]]; path[k] = pp[i]; }
    return off[n + 1] - off[n];
}
// linear steps and their run lengths (flat_graph.hpp; kernel_dp_band.hip), [L] each
int hlala_host_linear(const hlala::FlatGraph* F, uint32_t* lin_label, int32_t* lin_eid, uint8_t* lin_out, uint8_t* lin_in)
{
    memcpy(lin_label, F->lin_label.data(), F->lin_label.size() * 4); memcpy(lin_eid, F->lin_eid.data(), F->lin_eid.size() * 4);
    memcpy(lin_out, F->lin_out.data(), F->lin_out.size()); memcpy(lin_in, F->lin_in.data(), F->lin_in.size());
    return 0;
}
}
