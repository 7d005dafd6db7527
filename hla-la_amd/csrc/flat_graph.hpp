// flat_graph.hpp -- one-time host pass: PRG graph (graph.txt creation order) -> CSR arrays for HBM.
//
// Replaces, for the device path, the pointer structures the reference builds at start-up:
//   alignerBase::nodesPerLevel_ordered{,_rev}        mapper/aligner/alignerBase.cpp:27-37
//   Node::Outgoing_Edges / Incoming_Edges             Graph/Node.h:74-75  (std::set<Edge*> order)
//   Graph::computeGapEdgePaths                        Graph/Graph.cpp:347-476
//   processBAM::inGraphGapStretch                     mapper/processBAM.cpp:91-149
//   processBAM::graphLevel_2_underlyingSequencePositions   mapper/processBAM.cpp:4441-4456
// Canonical order everywhere = creation index.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/hlala_gpu.h"

namespace hlala {

struct FlatGraph {
    int32_t L = 0, N = 0, E = 0;
    // nodes are renumbered level-major (stable in creation index): rank z = id - level_off[level]
    std::vector<int32_t> level_off;      // [L+1]
    std::vector<int32_t> node_orig;      // [N] new id -> creation index
    std::vector<int32_t> node_new;       // [N] creation index -> new id
    std::vector<int32_t> node_level;     // [N] by new id
    // out-/in-edge CSR by new node id, entries in edge creation order
    std::vector<int32_t> out_off, in_off;        // [N+1]
    std::vector<int32_t> out_to, in_from;        // [E] new node ids
    std::vector<uint8_t> out_label, in_label;    // [E]
    std::vector<int32_t> out_eid, in_eid;        // [E] edge creation index
    std::vector<int32_t> edge_from_new, edge_to_new;   // [E] by creation index -> new node ids
    // completed gap-edge paths (completedGapEdgePaths order)
    std::vector<int32_t> path_first, path_last;  // [P] new node ids
    std::vector<int32_t> path_len;               // [P]
    std::vector<int64_t> path_off;               // [P+1] into path_edges
    std::vector<int32_t> path_edges;             // edge creation indices, first -> last
    // jump tables by new node id; entries ordered by the CREATION index of the other node
    std::vector<int32_t> jf_off, jb_off;         // [N+1]
    std::vector<int32_t> jf_node, jb_node;       // target node (new id)
    std::vector<int32_t> jf_path, jb_path;       // path index
    std::vector<int32_t> jf_lvl, jb_lvl;         // level of the target node (saves a dependent load on the device)
    // The jump tables the DEVICE walks (round 5): the entries above without the paths of ONE edge.  A gap path that consists of a single '_' edge p -> q is a no-op
    // for the extension DP: its candidate -- D[p] + 1 * S_graphGap into cell (level(q), y, q), extensionAligner.cpp:757-786 -- has the value of the candidate the
    // '_' edge itself pushed into the same cell a moment earlier (the non-affine sequence gap D[p] + S_graphGap, :738-752, in the same pass over p), and "first maximum
    // in push order" (Utilities.cpp:379-406) never picks the later of two equal candidates; the cell exists either way.  Dropping them changes no result, and a
    // call that only ever meets such paths (every single-level deletion of a haplotype is one) counts as jump-free (DpTinyJF).  keep_unit_jumps = true keeps all.
    std::vector<int32_t> djf_off, djb_off, djf_node, djb_node, djf_path, djb_path, djf_lvl, djb_lvl;
    // rank of a CSR edge / jump entry among the EARLIER entries of the same node that lead to the same target node.  It is all the extension DP needs to
    // order the candidates of one target cell (candidates of different edges of one frontier cell only compete when they reach the same node), so the push
    // index of a candidate holds this rank -- a few bits -- instead of the edge number, and a node's degree is not limited by the width of that field.
    std::vector<uint8_t> out_prank, in_prank, jf_prank, jb_prank;       // (jf_prank / jb_prank: over the device's tables djf_* / djb_*)
    // one word per in-edge (CSR order) for the re-threading DP of the seed projection (kernel_project.hip, chunked form): the lanes of a wavefront take
    // the in-edges of a level, not its nodes.  bits 0-8 rank of the from-node in its level; 9-14 the place, among the in-edges of the from-node's
    // level, of the from-node's LAST in-edge (the lane that holds the from-node's score when that level was solved in one slice of 64 edges); 15-17 the
    // edge's place among the in-edges of its target (0 .. 7); 18-25 label; 28 last in-edge of its target.
    // level_fast[l]: 0 = the level is solved node by node; else bits 0-1 = 1: edge-parallel in one slice (at most 64 in-edges), 2: in several; bits 2-4 = largest
    // in-degree - 1.  Edge-parallel needs: every node has one to eight in-edges, all from level l - 1, ranks below 512.
    std::vector<uint32_t> in_rec;                // [E]
    std::vector<uint8_t> level_fast;             // [L]
    // one 32-byte record per node and direction for the extension DP: {first CSR edge, degree | jumps << 16, target of edge 0,
    // target of edge 1, first jump-table entry, node of jump 0, level of jump 0, label 0 | label 1 << 8 | rank of edge 1 << 16}
    std::vector<int32_t> nrec_out, nrec_in;      // [8*N]
    // jfree_out[l]: number of consecutive levels l, l + 1, ... none of whose nodes has a forward gap-path jump (jfree_in: l, l - 1, ..., backward jumps), capped
    // at 255.  An extension DP that starts at level l and can reach at most r levels is known to meet no jump when jfree > r: such calls run in the
    // instantiation of the 16-lane class that is compiled without the early-cell machinery (kernel_dp.hip: DpTinyJF; a call that meets one anyway is re-run).
    std::vector<uint8_t> jfree_out, jfree_in;    // [L]
    // LINEAR steps (round 5, kernel_dp_band.hip): a step l -> l + 1 is linear when both levels hold exactly one node, one to four PARALLEL edges join them (the
    // SNPs of a merged backbone are such edges), no label is '_' and neither end has a gap-path jump along it.  lin_label[l] = the labels of those edges in CSR
    // (= creation) order, one per byte from the low byte up (0: not linear); lin_eid[l] = creation index of the first (-1: none).  lin_out[l] = number of
    // consecutive linear steps l -> l + 1, l + 1 -> l + 2, ...; lin_in[l] = steps l -> l - 1, l - 1 -> l - 2, ... (capped at 255).  Inside such a run the node rank
    // z is 0 everywhere, the frontier of an extension DP is a plain band of cells (level, read offset) and no cell is ever met twice: calls whose reach stays
    // inside a run take the register-resident anti-diagonal kernel instead of the hashed-frontier machine.
    std::vector<uint32_t> lin_label;                   // [L]
    std::vector<uint8_t> lin_out, lin_in;              // [L]
    std::vector<int32_t> lin_eid;                      // [L]
    // TRACK steps (round 6, kernel_dp_band2.hip): the neighbourhood of a gap stretch of the backbone is two nodes per level -- a base track and a '_' track, or two base
    // tracks until a SNP merges them -- with at most one gap-path jump across.  A step in the walking direction (out: level l -> l + 1 through the out-edges of the nodes
    // of l; in: level l -> l - 1 through the in-edges of the nodes of l) is a track step when both levels hold one or two nodes, every source node has one to four edges
    // and every label is one of A C G T N _.  trk_w_*[l] = its 64-bit word (0: not a track step): 16 bits per (source rank z', target rank z) pair at bits 16 (2 z' + z):
    // bit 0 some edge, 1 an edge with a real label, 2 a '_' edge, 3 the '_' edge precedes the first real edge of the pair in CSR order, 4..8 which of A C G T N label
    // the pair's real edges, 9..11 (pairs (z', 0)) the number of edges of source node z'.  trk_out / trk_in[l] = consecutive track steps from level l on (capped at 255).
    // trk_j_*[l]: the gap-path jumps of MORE than one edge that start at the nodes of level l in that direction (paths of one edge are no-ops, see above): 0 none;
    // bit 0 exactly one, and the kernel can take it (4 to 29 edges); bit 1 anything else (several, shorter, longer: a call that stands there fails over); bit 2 rank of
    // the source node, bit 3 rank of the target node, bits 8..15 edges of the path; trk_jp_*[l] = the path (index into path_len / path_off).
    std::vector<unsigned long long> trk_w_out, trk_w_in;         // [L]
    std::vector<uint8_t> trk_out, trk_in;              // [L]
    std::vector<uint32_t> trk_j_out, trk_j_in;         // [L]
    std::vector<int32_t> trk_jp_out, trk_jp_in;        // [L]
    std::vector<uint8_t> gap_stretch;            // [L-1]
    // level -> (sequence id, position) CSR, entries sorted by sequence id
    std::vector<int64_t> lp_off;                 // [L+1]
    std::vector<int32_t> lp_seqid, lp_pos;
    int32_t max_nodes_per_level = 0, max_out_degree = 0, max_in_degree = 0, max_jumps = 0, max_parallel = 0;       // max_parallel: largest rank + 1 above
};

// Returns "" on success, else the error text (graph invariants the reference asserts).
std::string flatten_graph(const hlala_graph_desc* g, const hlala_contigs_desc* c, FlatGraph& out, bool keep_unit_jumps = false);

}  // namespace hlala
