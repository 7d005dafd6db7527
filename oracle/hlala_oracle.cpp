/*
 * hlala_oracle.cpp -- CPU restatement of HLA*LA's read-to-PRG alignment hot path.
 *
 * THIS IS TEST INFRASTRUCTURE.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load it.  The product (libhlala_gpu.so) never links or calls it.
 *
 * Pinning status (see DESIGN.md "Oracle"):  the reference cannot be compiled in this image
 * (every translation unit on the path includes Boost and BamTools headers, which are absent,
 * and stand-ins are not allowed), so there is no oracle/_ref.  The restatement is pinned on
 * the known-answer material the reference itself carries for this path -- the
 * intervalsOverlap asserts (HLA-LA.cpp:94-102), the Phred round-trip table
 * (mapper/processBAM.cpp:4229-4239), the `--action testChainExtension` protocol
 * (HLA-LA.cpp:1733-1861: extended chain must re-spell the read) and the paranoid invariants
 * (verboseSeedChain.cpp:48-77, verboseSeedChain.h:282-315) -- and on glibc's own rand_r.
 * The DP cell values / tie-breaks themselves have NO golden vectors in the reference:
 * for those, parity is UNPINNED beyond the line-by-line restatement below.
 *
 * Every function cites the reference file:line it follows.  Data structures deliberately
 * keep the reference's std::map / std::set iteration orders (with node/edge creation index
 * standing in for heap-pointer order) so that tie-breaking is inherited
 * by construction rather than re-derived.
 */
#include "../include/hlala_gpu.h"

#include <algorithm>
#include <cassert>
#include <cfloat>
#include <climits>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <set>
#include <stdexcept>
#include <string>
#include <tuple>
#include <vector>
#include <omp.h>

namespace orc {

struct oracle_error : std::runtime_error {
    using std::runtime_error::runtime_error;
};
#define ORC_CHECK(cond, msg) do { if(!(cond)) throw oracle_error(std::string("oracle check failed: ") + msg + " [" #cond "]"); } while(0)

/* ---------------------------------------------------------------- Utilities.cpp helpers */

/* Utilities::intervalsOverlap, Utilities.cpp:168-176 */
static bool intervalsOverlap(int x1, int x2, int y1, int y2)
{
    return (x1 >= y1 && x1 <= y2) || (x2 >= y1 && x2 <= y2) || (y1 >= x1 && y1 <= x2) || (y2 >= x1 && y2 <= x2);
}

/* Utilities::PCorrectToPhred, Utilities.cpp:178-203 */
static unsigned char PCorrectToPhred(double PCorrect)
{
    double pWrong = 1 - PCorrect;
    if(pWrong == 0) pWrong = 1e-100;
    double phred1 = -10.0 * log10(pWrong);
    if((phred1 + 33) > 255) phred1 = 255 - 33;
    int r = (int)round(phred1 + 33);
    return (unsigned char)r;
}

/* Utilities::PhredToPCorrect, Utilities.cpp:357-377 */
static double PhredToPCorrect(unsigned char q)
{
    if(q == 0) return -1;
    int illuminaPhred = (int)q - 33;
    double log10_pWrong = (double)illuminaPhred / (double)-10;
    double pWrong = exp(log(10) * log10_pWrong);
    return 1 - pWrong;
}

/* Utilities::findVectorMax / findVectorMaxP_nonCritical, Utilities.cpp:309-323, 379-406:
 * only a strictly greater element replaces the maximum, so the FIRST maximum wins and the RNG
 * branch (iMaxs.size() > 1) is unreachable. */
static std::pair<double, unsigned> firstMax(const std::vector<double>& v)
{
    double mx = 0; unsigned iMax = 0;
    for(unsigned i = 0; i < v.size(); i++)
        if(i == 0 || v[i] > mx) { mx = v[i]; iMax = i; }
    return {mx, iMax};
}

/* Utilities::seq_reverse_complement's per-character map, Utilities.cpp:1170-1200 */
static char complementChar(char c)
{
    switch(c) {
        case 'A': return 'T'; case 'C': return 'G'; case 'G': return 'C'; case 'T': return 'A';
        case 'a': return 't'; case 'c': return 'g'; case 'g': return 'c'; case 't': return 'a';
        default: return c;
    }
}

/* boost::math::pdf(normal_distribution, x), boost/math/distributions/normal.hpp (Boost >= 1.59,
 * un-vendored, README.md:42): exponent = -(x-mean)^2 / (2 sd^2); exp(exponent) / (sd*sqrt(2 pi)).
 * Call sites processBAM.cpp:2343, 3446.  Parity with a real Boost build is unpinned. */
static double normal_pdf(double mean, double sd, double x)
{
    double exponent = x - mean;
    exponent *= -exponent;
    exponent /= 2 * sd * sd;
    double result = exp(exponent);
    result /= sd * sqrt(2 * 3.141592653589793238462643383279502884);
    return result;
}

/* --------------------------------------------------------------------------- the graph */

struct Graph {
    int L = 0;
    std::vector<int> node_level;
    std::vector<int> efrom, eto;
    std::vector<unsigned char> elabel;
    std::vector<std::vector<int>> level_nodes;   /* alignerBase::nodesPerLevel_ordered, alignerBase.cpp:27-37 */
    std::vector<int> node_rank;                  /* nodesPerLevel_ordered_rev                                 */
    std::vector<std::vector<int>> out_e, in_e;   /* Node::Outgoing_Edges / Incoming_Edges in set order        */
    std::vector<std::vector<int>> paths;         /* Graph::completedGapEdgePaths                              */
    std::map<int, std::map<int, int>> jump_fwd;  /* gapEdgePaths_connectedNodes_forwards: first -> last -> path */
    std::map<int, std::map<int, int>> jump_bwd;  /* gapEdgePaths_connectedNodes_backwards                      */
    std::vector<unsigned char> inGraphGapStretch;/* processBAM::inGraphGapStretch                              */

    void build(const hlala_graph_desc* d)
    {
        L = d->n_levels;
        ORC_CHECK(L > 1, "graph needs > 1 level");
        node_level.assign(d->node_level, d->node_level + d->n_nodes);
        efrom.assign(d->edge_from, d->edge_from + d->n_edges);
        eto.assign(d->edge_to, d->edge_to + d->n_edges);
        elabel.assign(d->edge_label, d->edge_label + d->n_edges);
        level_nodes.assign(L, {});
        node_rank.assign(d->n_nodes, -1);
        for(int n = 0; n < d->n_nodes; n++) {
            ORC_CHECK(node_level[n] >= 0 && node_level[n] < L, "node level range");
            node_rank[n] = (int)level_nodes[node_level[n]].size();
            level_nodes[node_level[n]].push_back(n);
        }
        out_e.assign(d->n_nodes, {});
        in_e.assign(d->n_nodes, {});
        for(int e = 0; e < d->n_edges; e++) {
            ORC_CHECK(node_level[eto[e]] == node_level[efrom[e]] + 1, "edge must connect consecutive levels");
            out_e[efrom[e]].push_back(e);
            in_e[eto[e]].push_back(e);
        }
        computeGapEdgePaths();
        computeGapStretches();
    }

    /* Graph::computeGapEdgePaths, Graph/Graph.cpp:347-476 */
    void computeGapEdgePaths()
    {
        std::map<int, std::map<int, std::vector<int>>> runningPaths;   /* node at this level -> origin node -> edge path */
        for(int lI = 0; lI < L; lI++) {
            std::map<int, std::map<int, std::vector<int>>> runningPaths_nextLevel;
            std::set<int> seen_gap_edge;
            for(auto& nodeIt : runningPaths) {
                int thisLevelNode = nodeIt.first;
                int non_gap_edges = 0;
                for(int e : out_e[thisLevelNode]) {
                    if(elabel[e] == '_') {
                        seen_gap_edge.insert(e);
                        int targetNode = eto[e];
                        for(auto& fromNodeIt : nodeIt.second) {
                            int fromNode = fromNodeIt.first;
                            if(runningPaths_nextLevel.count(targetNode) == 0 || runningPaths_nextLevel.at(targetNode).count(fromNode) == 0) {
                                std::vector<int> p = fromNodeIt.second;
                                p.push_back(e);
                                runningPaths_nextLevel[targetNode][fromNode] = p;
                            }
                        }
                    } else {
                        non_gap_edges++;
                    }
                }
                if(non_gap_edges != 0 || lI == L - 1)
                    for(auto& fromNodeIt : nodeIt.second)
                        paths.push_back(fromNodeIt.second);
            }
            /* Graph::getEdgesEmanatingFromLevel (Graph.cpp:556-565): a std::set<Edge*> = edge creation order */
            std::set<int> edges_thisLevel;
            for(int n : level_nodes[lI]) edges_thisLevel.insert(out_e[n].begin(), out_e[n].end());
            for(int e : edges_thisLevel) {
                if(elabel[e] == '_' && seen_gap_edge.count(e) == 0) {
                    int fromNode = efrom[e], targetNode = eto[e];
                    if(runningPaths_nextLevel.count(targetNode) == 0 || runningPaths_nextLevel.at(targetNode).count(fromNode) == 0)
                        runningPaths_nextLevel[targetNode][fromNode] = std::vector<int>{e};
                }
            }
            runningPaths = runningPaths_nextLevel;
        }
        for(int pathI = 0; pathI < (int)paths.size(); pathI++) {
            int firstNode = efrom[paths[pathI].front()];
            int lastNode = eto[paths[pathI].back()];
            ORC_CHECK(jump_fwd.count(firstNode) == 0 || jump_fwd.at(firstNode).count(lastNode) == 0, "duplicate gap path");
            jump_fwd[firstNode][lastNode] = pathI;
            jump_bwd[lastNode][firstNode] = pathI;
        }
    }

    /* processBAM::processBAM gap-stretch scan, mapper/processBAM.cpp:91-149 */
    void computeGapStretches()
    {
        inGraphGapStretch.assign(L - 1, 0);
        const int gapStretchMinimumLength = 3;
        auto addGapStretch = [&](int a, int b) {
            if(b - a + 1 >= gapStretchMinimumLength)
                for(int l = a; l <= b; l++) inGraphGapStretch[l] = 1;
        };
        int stretchStart = -1;
        for(int lI = 0; lI < L - 1; lI++) {
            bool haveGapEdge = false;
            for(int n : level_nodes[lI]) for(int e : out_e[n]) if(elabel[e] == '_') haveGapEdge = true;
            if(haveGapEdge) { if(stretchStart == -1) stretchStart = lI; }
            else if(stretchStart != -1) { addGapStretch(stretchStart, lI - 1); stretchStart = -1; }
        }
        if(stretchStart != -1) addGapStretch(stretchStart, L - 2);
    }
};

/* mapper::reads::verboseSeedChain, mapper/reads/verboseSeedChain.h:22-50 */
struct Chain {
    int sequence_begin = -1, sequence_end = -1;
    bool reverse = false;
    int removed_columns_noGap_restriction = -1;
    double improvement_through_bt = -1;
    std::vector<int> levels;       /* graph_aligned_levels */
    std::vector<int> edges;        /* graph_aligned_edges (creation index, -1 = null) */
    std::string graph_aligned, sequence_aligned;
    std::vector<unsigned char> is_from_BWAseed;
    double mapQ = 0;
    std::string mapQ_perPosition;
    double ll = 0;
    int dp_iters[2] = {0, 0};
    int dp_score[2] = {INT_MIN, INT_MIN};

    size_t size() const { return levels.size(); }
    /* verboseSeedChain::alignment_firstLevel / lastLevel, verboseSeedChain.h:118-190 */
    int firstLevel() const { for(int l : levels) if(l != -1) return l; return -1; }
    int lastLevel() const { for(int i = (int)levels.size() - 1; i >= 0; i--) if(levels[i] != -1) return levels[i]; return -1; }
    std::vector<int> firstLevels(int n) const { std::vector<int> r; for(int l : levels) if(l != -1) { r.push_back(l); if((int)r.size() >= n) break; } return r; }
    std::vector<int> lastLevels(int n) const { std::vector<int> r; for(int i = (int)levels.size() - 1; i >= 0; i--) if(levels[i] != -1) { r.push_back(levels[i]); if((int)r.size() >= n) break; } return r; }

    /* verboseSeedChain::checkChainConcordanceWithSequence, verboseSeedChain.cpp:48-77 */
    void checkConcordance(const std::string& sequence) const
    {
        ORC_CHECK(edges.size() == levels.size() && graph_aligned.size() == levels.size() && sequence_aligned.size() == levels.size(), "chain array sizes");
        ORC_CHECK(sequence_begin <= sequence_end && sequence_begin >= 0 && sequence_end < (int)sequence.size(), "chain sequence range");
        std::string noGaps;
        for(char c : sequence_aligned) if(c != '_') noGaps.push_back(c);
        ORC_CHECK(sequence.substr(sequence_begin, sequence_end - sequence_begin + 1) == noGaps, "chain concordance with sequence");
    }
    /* verboseSeedChain::checkLevelContiguity, verboseSeedChain.h:282-315 */
    void checkLevelContiguity() const
    {
        int last = -1;
        for(int l : levels) if(l != -1) { ORC_CHECK(last == -1 || last + 1 == l, "level contiguity"); last = l; }
    }
    /* verboseSeedChain::extendWithOtherSeedChain, verboseSeedChain.cpp:23-46 */
    void extendWith(const Chain& o, bool left)
    {
        if(left) {
            ORC_CHECK(o.sequence_end + 1 == sequence_begin, "left extension adjacency");
            sequence_begin = o.sequence_begin;
            levels.insert(levels.begin(), o.levels.begin(), o.levels.end());
            edges.insert(edges.begin(), o.edges.begin(), o.edges.end());
            graph_aligned.insert(graph_aligned.begin(), o.graph_aligned.begin(), o.graph_aligned.end());
            sequence_aligned.insert(sequence_aligned.begin(), o.sequence_aligned.begin(), o.sequence_aligned.end());
            is_from_BWAseed.insert(is_from_BWAseed.begin(), o.is_from_BWAseed.begin(), o.is_from_BWAseed.end());
        } else {
            ORC_CHECK(o.sequence_begin == sequence_end + 1, "right extension adjacency");
            sequence_end = o.sequence_end;
            levels.insert(levels.end(), o.levels.begin(), o.levels.end());
            edges.insert(edges.end(), o.edges.begin(), o.edges.end());
            graph_aligned.insert(graph_aligned.end(), o.graph_aligned.begin(), o.graph_aligned.end());
            sequence_aligned.insert(sequence_aligned.end(), o.sequence_aligned.begin(), o.sequence_aligned.end());
            is_from_BWAseed.insert(is_from_BWAseed.end(), o.is_from_BWAseed.begin(), o.is_from_BWAseed.end());
        }
    }
    /* verboseSeedChain::extendToFullSequenceLength, verboseSeedChain.cpp:79-136 */
    void extendToFull(const std::string& sequence)
    {
        int missing_left = sequence_begin;
        int missing_right = (int)sequence.size() - sequence_end - 1;
        if(missing_left) {
            edges.insert(edges.begin(), missing_left, -1);
            levels.insert(levels.begin(), missing_left, -1);
            graph_aligned.insert(graph_aligned.begin(), missing_left, '_');
            sequence_aligned.insert(0, sequence.substr(0, missing_left));
            is_from_BWAseed.insert(is_from_BWAseed.begin(), missing_left, 0);
            sequence_begin = 0;
        }
        if(missing_right) {
            edges.insert(edges.end(), missing_right, -1);
            levels.insert(levels.end(), missing_right, -1);
            graph_aligned.insert(graph_aligned.end(), missing_right, '_');
            sequence_aligned.append(sequence.substr(sequence.size() - missing_right, missing_right));
            is_from_BWAseed.insert(is_from_BWAseed.end(), missing_right, 0);
            sequence_end = (int)sequence.size() - 1;
        }
        checkConcordance(sequence);
    }
};

/* ------------------------------------------------------------------ extension aligner */

/* Work counters of the calling thread (the all-cores variant of orc_align_batch runs pairs on several threads; the aligner itself is
 * read-only): cells / iterations / calls / edges, and the largest per-call sizes seen (capacity planning of the GPU classes). */
struct DpStats {
    long long cells = 0, iters = 0, calls = 0, edges = 0;
    long long max_frontier = 0, max_targets = 0, max_kept_cells = 0, max_completed = 0;
    long long ovf_calls = 0, ovf_first_iter = 0, ovf_total_iter = 0;   /* calls whose frontier passes 16 cells or whose target set passes 24: the iteration where it first happens, and their iterations in all */
    long long hist_frontier[16] = {0}, hist_targets[16] = {0};   /* per DP call: bucket ceil(log2) of its widest frontier / target set (tools: capacity classes) */
    bool h4_hit = false;
    void add(const DpStats& o) {
        cells += o.cells; iters += o.iters; calls += o.calls; edges += o.edges;
        max_frontier = std::max(max_frontier, o.max_frontier); max_targets = std::max(max_targets, o.max_targets);
        for(int i = 0; i < 16; i++) { hist_frontier[i] += o.hist_frontier[i]; hist_targets[i] += o.hist_targets[i]; }
        ovf_calls += o.ovf_calls; ovf_first_iter += o.ovf_first_iter; ovf_total_iter += o.ovf_total_iter;
        max_kept_cells = std::max(max_kept_cells, o.max_kept_cells); max_completed = std::max(max_completed, o.max_completed); h4_hit = h4_hit || o.h4_hit;
    }
};
static thread_local DpStats t_stats;

struct Aligner {
    const Graph* g;
    /* alignerBase::alignerBase, mapper/aligner/alignerBase.cpp:19-25 */
    double S_match = 2, S_mismatch = -5, S_gap = -2, S_graphGap = 0, S_openGap = -4, S_extendGap = -2;

    explicit Aligner(const Graph* g_) : g(g_) {}

    /* backtraceStep_affine, alignerBase.h:36-51.  edge >= 0: graph edge; edge == -1: none;
     * edge <= -2: pseudo edge of gap path (-2 - edge). */
    struct BT { int x = -1, y = -1, z = -1, src = -1, edge = -1; };
    struct Cell { double D, GG, SG; };
    struct CellBT { BT D, GG, SG; };
    struct Alt { std::vector<double> D, GG, SG; std::vector<BT> bD, bGG, bSG; };
    typedef std::tuple<int, int, int> Key;

    /* alignerBase::_graph_get_{next,previous}_z_values_and_edges, alignerBase.cpp:149-195 */
    std::vector<std::pair<int, int>> neighbours(int x, int z, bool fwd) const
    {
        std::vector<std::pair<int, int>> r;
        int n = g->level_nodes.at(x).at(z);
        if(fwd) for(int e : g->out_e[n]) r.push_back({g->node_rank[g->eto[e]], e});
        else    for(int e : g->in_e[n])  r.push_back({g->node_rank[g->efrom[e]], e});
        return r;
    }
    /* extensionAligner::_graph_get_jump{Next,Previous}_x_and_z_values_and_edges, extensionAligner.cpp:2694-2742 */
    std::vector<std::tuple<int, int, int>> jumps(int x, int z, bool fwd) const
    {
        std::vector<std::tuple<int, int, int>> r;   /* (level, z, pathIdx) */
        int n = g->level_nodes.at(x).at(z);
        const auto& tbl = fwd ? g->jump_fwd : g->jump_bwd;
        auto it = tbl.find(n);
        if(it != tbl.end())
            for(auto& t : it->second)
                r.push_back(std::make_tuple(g->node_level[t.first], g->node_rank[t.first], t.second));
        return r;
    }

    struct Ext { bool have = false; Chain chain; int iters = 0; int score = INT_MIN; };
    /* test instrumentation (tools/band2: the CPU model of a GPU kernel is run on the arguments of every DP call and compared with its result): no part of the restatement */
    struct DpObserver { virtual ~DpObserver() {} virtual void seen(const Aligner& A, const std::string& sequence, int start_sequence, int startLevel, int startZ, bool fwd, unsigned int seedBefore, const Ext& result, long long cells, long long edges) = 0; };
    static DpObserver*& observer() { static thread_local DpObserver* o = nullptr; return o; }

    /* extensionAligner::fullNeedleman_diagonal_extension_gapJumper, extensionAligner.cpp:335-1556,
     * in the only configuration the path uses (extendSeedChain, :229-241 and :281-293):
     * returnGlobalScore = false, preferSequenceCompleAlignments = true, empty blockedPathsTable. */
    Ext dp(const std::string& sequence, int start_sequence, int startLevel_graph, int startZ_graph,
           int maxLevel_graph, int maxPosition_sequence, int diagonal_stop_threshold, bool directionPositive,
           unsigned int* rng_seed)
    {
        t_stats.calls++;
        const unsigned int seedBefore_ = *rng_seed; const long long cells0_ = t_stats.cells, edges0_ = t_stats.edges;
        long long callMaxF = 0, callMaxT = 0, callFirstOvf = -1;
        const double minusInfinity = -1 * DBL_MAX;                                   /* :363 */
        std::map<Key, Cell> scores;                                                    /* :396 */
        std::map<Key, CellBT> scores_backtrace;                                        /* :397 */
        std::vector<Key> m1_diagonal, m2_diagonal;

        int levels = g->L;
        int sequenceLength = (int)sequence.size();
        int diagonals = sequenceLength + levels - 1;                                  /* :431 */
        int max_levelI = levels - 1, max_seqI = sequenceLength, min_levelI = 0, min_seqI = 0;
        if(maxLevel_graph != -1) { if(directionPositive) max_levelI = maxLevel_graph; else min_levelI = maxLevel_graph; }
        if(maxPosition_sequence != -1) { if(directionPositive) max_seqI = maxPosition_sequence; else min_seqI = maxPosition_sequence; }
        ORC_CHECK(startLevel_graph >= min_levelI && startLevel_graph <= max_levelI, "DP start level");      /* :465-478 */
        ORC_CHECK(start_sequence >= min_seqI && start_sequence <= max_seqI, "DP start seq");
        ORC_CHECK(max_levelI > min_levelI && max_seqI > min_seqI, "DP extent");

        double currentMaximum = 0;                                                    /* :480 */
        std::vector<Key> currentMaxima_coordinates;
        int statesPerLevel0 = (int)g->level_nodes.at(startLevel_graph).size();
        ORC_CHECK(startZ_graph >= 0 && startZ_graph < statesPerLevel0, "DP start z");
        const int threshold_for_filtering = 15;                                       /* :489 */
        const int maximum_steps_nonIncrease = 40;                                     /* :490 */
        std::set<std::string> achieved_complete_sequence_alignments;                  /* :493 */

        for(int stateI = 0; stateI < statesPerLevel0; stateI++) {                     /* :495-519 */
            Key k(startLevel_graph, start_sequence, stateI);
            Cell c;
            c.D = (stateI == startZ_graph) ? 0 : minusInfinity;
            c.GG = minusInfinity; c.SG = minusInfinity;
            scores[k] = c;
            if(stateI == startZ_graph) {
                scores_backtrace[k] = CellBT();
                m1_diagonal.push_back(k);
                currentMaxima_coordinates.push_back(k);
            }
        }

        int lastMaximumIncrease_at_diagonalI = 0;
        int itersRun = 0;
        for(int diagonalI = 1; diagonalI <= diagonals; diagonalI++) {                 /* :531 */
            if((diagonalI - lastMaximumIncrease_at_diagonalI) > maximum_steps_nonIncrease) break;   /* :553 */
            itersRun++;
            std::map<Key, Alt> thisDiagonal;

            /* extend from m-2 diagonal, :565-607 */
            for(const Key& p : m2_diagonal) {
                int px = std::get<0>(p), py = std::get<1>(p), pz = std::get<2>(p);
                int nx = px + (directionPositive ? 1 : -1), ny = py + (directionPositive ? 1 : -1);
                if(nx > max_levelI || ny > max_seqI) continue;
                if(nx < min_levelI || ny < min_seqI) continue;
                char sequenceEmission = directionPositive ? sequence.at(py) : sequence.at(py - 1);
                auto nextZs = neighbours(px, pz, directionPositive);
                ORC_CHECK(nextZs.size() > 0, "node without neighbours");
                for(auto& zj : nextZs) {
                    t_stats.edges++;
                    char edgeEmission = (char)g->elabel[zj.second];
                    double s = scores.at(p).D + ((edgeEmission == sequenceEmission) ? S_match : S_mismatch);
                    BT b; b.x = px; b.y = py; b.z = pz; b.edge = zj.second; b.src = 0;
                    Alt& a = thisDiagonal[Key(nx, ny, zj.first)];
                    a.D.push_back(s); a.bD.push_back(b);
                }
            }

            /* extend from m-1 diagonal, :613-787 */
            for(const Key& p : m1_diagonal) {
                int px = std::get<0>(p), py = std::get<1>(p), pz = std::get<2>(p);
                const Cell& pc = scores.at(p);
                /* gap in graph, :621-661 */
                {
                    int nx = px, ny = py + (directionPositive ? 1 : -1);
                    if((directionPositive && nx <= max_levelI && ny <= max_seqI) || (!directionPositive && nx >= min_levelI && ny >= min_seqI)) {
                        Alt& a = thisDiagonal[Key(nx, ny, pz)];
                        BT bo; bo.x = px; bo.y = py; bo.z = pz; bo.edge = -1; bo.src = 0;
                        a.GG.push_back(pc.D + S_openGap + S_extendGap); a.bGG.push_back(bo);
                        BT be = bo; be.src = 1;
                        a.GG.push_back(pc.GG + S_extendGap); a.bGG.push_back(be);
                    }
                }
                /* gap in sequence, :664-754 */
                {
                    int nx = px + (directionPositive ? 1 : -1), ny = py;
                    if((directionPositive && nx <= max_levelI && ny <= max_seqI) || (!directionPositive && nx >= min_levelI && ny >= min_seqI)) {
                        auto nextZs = neighbours(px, pz, directionPositive);
                        ORC_CHECK(nextZs.size() > 0, "node without neighbours");
                        for(auto& zj : nextZs) {
                            t_stats.edges++;
                            bool gapEdge = (g->elabel[zj.second] == '_');
                            Alt& a = thisDiagonal[Key(nx, ny, zj.first)];
                            double s_open = pc.D + S_openGap + S_extendGap;
                            if(gapEdge) s_open = minusInfinity;
                            BT bo; bo.x = px; bo.y = py; bo.z = pz; bo.edge = zj.second; bo.src = 0;
                            a.SG.push_back(s_open); a.bSG.push_back(bo);
                            double s_ext = pc.SG + S_extendGap;
                            if(gapEdge) s_ext = (pc.SG == minusInfinity) ? minusInfinity : pc.SG + S_graphGap;
                            BT be = bo; be.src = 2;
                            a.SG.push_back(s_ext); a.bSG.push_back(be);
                            if(gapEdge) {                                       /* non-affine sequence gap, :738-752 */
                                BT bn = bo; bn.src = 0;
                                a.D.push_back(pc.D + S_graphGap); a.bD.push_back(bn);
                            }
                        }
                    }
                }
                /* gap in sequence, jump, :757-786 */
                for(auto& j : jumps(px, pz, directionPositive)) {
                    int jx = std::get<0>(j), jy = py, jz = std::get<1>(j), pathI = std::get<2>(j);
                    if((directionPositive && jx <= max_levelI && jy <= max_seqI) || (!directionPositive && jx >= min_levelI && jy >= min_seqI)) {
                        int jump_length = (int)g->paths.at(pathI).size();
                        BT b; b.x = px; b.y = py; b.z = pz; b.edge = -2 - pathI; b.src = 0;
                        Alt& a = thisDiagonal[Key(jx, jy, jz)];
                        a.D.push_back(pc.D + (jump_length * S_graphGap)); a.bD.push_back(b);
                    }
                }
            }

            /* call maxima for this diagonal, :794-1073 */
            std::vector<Key> m_thisDiagonal;
            for(auto& diagIt : thisDiagonal) {
                t_stats.cells++;
                const Key& k = diagIt.first;
                int levelI = std::get<0>(k), seqI = std::get<1>(k), stateI = std::get<2>(k);
                Alt& a = diagIt.second;
                double selGG = minusInfinity, selSG = minusInfinity;
                BT stepGG, stepSG;
                if(a.GG.size()) { auto m = firstMax(a.GG); selGG = m.first; stepGG = a.bGG[m.second]; }
                if(a.SG.size()) { auto m = firstMax(a.SG); selSG = m.first; stepSG = a.bSG[m.second]; }
                BT fromGG; fromGG.x = levelI; fromGG.y = seqI; fromGG.z = stateI; fromGG.src = 1; fromGG.edge = -1;
                a.D.push_back(selGG); a.bD.push_back(fromGG);
                BT fromSG = fromGG; fromSG.src = 2;
                a.D.push_back(selSG); a.bD.push_back(fromSG);
                auto maxD = firstMax(a.D);
                /* blockedPathsTable is empty on this path (extensionAligner.cpp:210-218): blockOutCell never fires */
                if(maxD.first >= diagonal_stop_threshold) {                              /* :949 */
                    bool newEntry = (scores.count(k) == 0);
                    bool overwrittenEntry = false;
                    if(newEntry || scores.at(k).D < maxD.first) {
                        overwrittenEntry = !newEntry;
                        scores[k].D = maxD.first; scores_backtrace[k].D = a.bD[maxD.second];
                    }
                    if(newEntry || scores.at(k).GG < selGG) {
                        overwrittenEntry = !newEntry;
                        scores[k].GG = selGG; scores_backtrace[k].GG = stepGG;
                    }
                    if(newEntry || scores.at(k).SG < selSG) {
                        overwrittenEntry = !newEntry;
                        scores[k].SG = selSG; scores_backtrace[k].SG = stepSG;
                    }
                    if((directionPositive && seqI == max_seqI) || (!directionPositive && seqI == min_seqI))     /* :982-999 */
                        achieved_complete_sequence_alignments.insert(std::to_string(levelI) + "/" + std::to_string(stateI));
                    m_thisDiagonal.push_back(k);

                    BT oneRealStepBackwards = scores_backtrace[k].D;                          /* :1007-1041 */
                    while(oneRealStepBackwards.x == levelI && oneRealStepBackwards.y == seqI) {
                        Key kk(oneRealStepBackwards.x, oneRealStepBackwards.y, oneRealStepBackwards.z);
                        ORC_CHECK(oneRealStepBackwards.src != 0, "same-cell hop from D");
                        if(oneRealStepBackwards.src == 1) oneRealStepBackwards = scores_backtrace[kk].GG;
                        else oneRealStepBackwards = scores_backtrace[kk].SG;
                    }
                    double prevD;
                    {
                        Key kk(oneRealStepBackwards.x, oneRealStepBackwards.y, oneRealStepBackwards.z);
                        const Cell& pc2 = scores[kk];
                        prevD = (oneRealStepBackwards.src == 0) ? pc2.D : (oneRealStepBackwards.src == 1 ? pc2.GG : pc2.SG);
                    }
                    int previousScore;     /* `int previousScore = <double>`: x86 cvttsd2si gives INT_MIN out of range */
                    if(prevD <= (double)INT_MIN || prevD >= (double)INT_MAX) { previousScore = INT_MIN; t_stats.h4_hit = true; }
                    else previousScore = (int)prevD;
                    int scoreDifference = (int)(maxD.first - previousScore);
                    if(maxD.first == currentMaximum) {
                        if(scoreDifference != 0) {
                            currentMaxima_coordinates.push_back(k);
                            lastMaximumIncrease_at_diagonalI = diagonalI;
                        }
                    } else if(maxD.first > currentMaximum) {
                        currentMaximum = maxD.first;
                        currentMaxima_coordinates.clear();
                        currentMaxima_coordinates.push_back(k);
                        lastMaximumIncrease_at_diagonalI = diagonalI;
                    }
                    if(overwrittenEntry) lastMaximumIncrease_at_diagonalI = diagonalI;
                }
            }

            /* filtering, :1076-1102 */
            if(m_thisDiagonal.size() > 0) {
                double mx = 0;
                for(size_t i = 0; i < m_thisDiagonal.size(); i++) {
                    double S = scores.at(m_thisDiagonal[i]).D;
                    if(i == 0 || mx < S) mx = S;
                }
                std::vector<Key> filtered;
                for(const Key& c : m_thisDiagonal)
                    if((mx - scores.at(c).D) <= threshold_for_filtering) filtered.push_back(c);
                m_thisDiagonal = filtered;
            }
            if((long long)thisDiagonal.size() > t_stats.max_targets) t_stats.max_targets = (long long)thisDiagonal.size();
            if((long long)m_thisDiagonal.size() > t_stats.max_frontier) t_stats.max_frontier = (long long)m_thisDiagonal.size();
            if(callFirstOvf < 0 && ((long long)thisDiagonal.size() > 24 || (long long)m_thisDiagonal.size() > 16)) callFirstOvf = itersRun;
            if((long long)thisDiagonal.size() > callMaxT) callMaxT = (long long)thisDiagonal.size();
            if((long long)m_thisDiagonal.size() > callMaxF) callMaxF = (long long)m_thisDiagonal.size();
            m2_diagonal = m1_diagonal;                                               /* :1104-1105 */
            m1_diagonal = m_thisDiagonal;
        }
        t_stats.iters += itersRun;
        if(callFirstOvf >= 0) { t_stats.ovf_calls++; t_stats.ovf_first_iter += callFirstOvf; t_stats.ovf_total_iter += itersRun; }
        { auto bucket = [](long long v) { int b = 0; while((1ll << b) < v && b < 15) b++; return b; }; t_stats.hist_frontier[bucket(callMaxF)]++; t_stats.hist_targets[bucket(callMaxT)]++; }
        if((long long)scores_backtrace.size() > t_stats.max_kept_cells) t_stats.max_kept_cells = (long long)scores_backtrace.size();
        if((long long)achieved_complete_sequence_alignments.size() > t_stats.max_completed) t_stats.max_completed = (long long)achieved_complete_sequence_alignments.size();

        /* backtraceFrom, :1109-1354 */
        auto backtraceFrom = [&](int start_x, int start_y, int start_z, double StartScore) -> Ext {
            int bx = start_x, by = start_y, bz = start_z, bm = 0;
            std::string rec_graph, rec_seq;
            std::vector<int> rec_levels, used_edges;
            std::vector<Key> coords;
            coords.push_back(Key(start_x, start_y, start_z));
            while(bx != startLevel_graph || by != start_sequence) {
                const CellBT& cb = scores_backtrace.at(Key(bx, by, bz));
                BT step = (bm == 0) ? cb.D : (bm == 1 ? cb.GG : cb.SG);
                char sequenceEmission = 0;
                if(by >= 1 && directionPositive) sequenceEmission = sequence.at(by - 1);
                if(by < max_seqI && !directionPositive) sequenceEmission = sequence.at(by);
                int nx = step.x, ny = step.y, nz = step.z, nm = step.src;
                bool dontAdd = false;
                if(step.edge > -2) {                                   /* not a pseudo edge */
                    char edgeEmission = step.edge >= 0 ? (char)g->elabel[step.edge] : 0;
                    int dirx = directionPositive ? -1 : 1;
                    int lvl = directionPositive ? bx - 1 : bx;
                    if(nx == bx + dirx && ny == by + dirx) {           /* match or mismatch */
                        ORC_CHECK(step.edge >= 0, "match step without edge");
                        rec_graph.push_back(edgeEmission); rec_levels.push_back(lvl); rec_seq.push_back(sequenceEmission); used_edges.push_back(step.edge);
                    } else if(nx == bx && ny == by + dirx) {           /* gap in graph */
                        rec_graph.push_back('_'); rec_levels.push_back(-1); rec_seq.push_back(sequenceEmission); used_edges.push_back(-1);
                    } else if(nx == bx + dirx && ny == by) {           /* gap in sequence */
                        ORC_CHECK(step.edge >= 0, "sequence-gap step without edge");
                        rec_graph.push_back(edgeEmission); rec_levels.push_back(lvl); rec_seq.push_back('_'); used_edges.push_back(step.edge);
                    } else {
                        dontAdd = true;
                        ORC_CHECK(bx == nx && by == ny && bz == nz && nm != bm, "matrix hop");
                    }
                } else {                                              /* gap-path jump, :1282-1307 */
                    std::vector<int> edgePath = g->paths.at(-2 - step.edge);
                    std::vector<int> graph_levels;
                    for(int e : edgePath) graph_levels.push_back(g->node_level[g->efrom[e]]);
                    if(directionPositive) { std::reverse(graph_levels.begin(), graph_levels.end()); std::reverse(edgePath.begin(), edgePath.end()); }
                    rec_levels.insert(rec_levels.end(), graph_levels.begin(), graph_levels.end());
                    rec_graph.append(edgePath.size(), '_');
                    rec_seq.append(edgePath.size(), '_');
                    used_edges.insert(used_edges.end(), edgePath.begin(), edgePath.end());
                }
                bx = nx; by = ny; bz = nz; bm = nm;
                if(!dontAdd) coords.push_back(Key(nx, ny, nz));
            }
            if(directionPositive) {
                std::reverse(rec_graph.begin(), rec_graph.end()); std::reverse(rec_levels.begin(), rec_levels.end());
                std::reverse(rec_seq.begin(), rec_seq.end()); std::reverse(used_edges.begin(), used_edges.end());
                std::reverse(coords.begin(), coords.end());
            }
            /* localExtension_pathDescription::toVerboseSeedChain, VirtualNWUnique.cpp:20-40 */
            Ext r; r.have = true;
            r.chain.edges = used_edges; r.chain.levels = rec_levels; r.chain.graph_aligned = rec_graph; r.chain.sequence_aligned = rec_seq;
            r.chain.sequence_begin = std::get<1>(coords.front());
            r.chain.sequence_end = std::get<1>(coords.back()) - 1;
            ORC_CHECK(r.chain.sequence_begin <= r.chain.sequence_end, "extension consumes no read base");
            r.chain.reverse = false;
            r.score = (int)StartScore;
            return r;
        };

        /* end cell, :1381-1517 */
        Ext result;
        {
            int coordinate_seqI = directionPositive ? max_seqI : min_seqI;
            std::vector<std::string> best; double maxScore = 0; bool first = true;
            for(const std::string& coordinates : achieved_complete_sequence_alignments) {
                size_t sl = coordinates.find('/');
                int cl = atoi(coordinates.substr(0, sl).c_str()), cs = atoi(coordinates.substr(sl + 1).c_str());
                double S = scores.at(Key(cl, coordinate_seqI, cs)).D;
                if(first || S > maxScore) { best.clear(); best.push_back(coordinates); maxScore = S; first = false; }
                else if(S == maxScore) best.push_back(coordinates);
            }
            if(best.size() > 0) {
                /* Utilities::randomNumber_nonCritical, Utilities.cpp:922-927 */
                int selectedIndex = rand_r(rng_seed) % (int)best.size();
                const std::string& coordinates = best.at(selectedIndex);
                size_t sl = coordinates.find('/');
                int cl = atoi(coordinates.substr(0, sl).c_str()), cs = atoi(coordinates.substr(sl + 1).c_str());
                result = backtraceFrom(cl, coordinate_seqI, cs, maxScore);
            } else if(currentMaximum > 0) {
                /* :1481-1497 backtraces from every maximum, but :1548-1552 returns copies of forReturn.at(0) only */
                for(const Key& c : currentMaxima_coordinates) {
                    if(scores.at(c).D != minusInfinity) {
                        result = backtraceFrom(std::get<0>(c), std::get<1>(c), std::get<2>(c), scores.at(c).D);
                        break;
                    }
                }
            }
        }
        result.iters = itersRun;
        if(observer()) observer()->seen(*this, sequence, start_sequence, startLevel_graph, startZ_graph, directionPositive, seedBefore_, result, t_stats.cells - cells0_, t_stats.edges - edges0_);
        return result;
    }

    /* extensionAligner::extendSeedChain, extensionAligner.cpp:186-333 */
    Chain extendSeedChain(const std::string& sequence, const Chain& seedChain, unsigned int seed_left, unsigned int seed_right)
    {
        ORC_CHECK(seedChain.sequence_begin <= seedChain.sequence_end && seedChain.sequence_begin >= 0, "seed range");
        ORC_CHECK(seedChain.sequence_end < (int)sequence.size(), "seed end");
        seedChain.checkConcordance(sequence);
        Chain forReturn = seedChain;
        ORC_CHECK(seedChain.edges.size() > 0, "empty seed chain");
        if(seedChain.sequence_begin != 0) {                                            /* left, :220-268 */
            int e0 = seedChain.edges.front();
            ORC_CHECK(e0 >= 0, "first seed column has no edge");
            int firstNode = g->efrom[e0];
            if(g->node_level[firstNode] > 0) {
                unsigned int s = seed_left;
                Ext x = dp(sequence, seedChain.sequence_begin, g->node_level[firstNode], g->node_rank[firstNode], 0, 0, -16, false, &s);
                forReturn.dp_iters[0] = x.iters;
                if(x.have) {
                    x.chain.reverse = seedChain.reverse;
                    x.chain.checkConcordance(sequence);
                    x.chain.is_from_BWAseed.assign(x.chain.size(), 0);
                    forReturn.extendWith(x.chain, true);
                    forReturn.dp_score[0] = x.score;
                }
            }
        }
        if(seedChain.sequence_end != (int)sequence.size() - 1) {                       /* right, :271-319 */
            int e1 = seedChain.edges.back();
            ORC_CHECK(e1 >= 0, "last seed column has no edge");
            int lastNode = g->eto[e1];
            if(g->node_level[lastNode] < g->L - 1) {
                unsigned int s = seed_right;
                Ext x = dp(sequence, seedChain.sequence_end + 1, g->node_level[lastNode], g->node_rank[lastNode], g->L - 1, (int)sequence.size(), -16, true, &s);
                forReturn.dp_iters[1] = x.iters;
                if(x.have) {
                    x.chain.reverse = seedChain.reverse;
                    x.chain.checkConcordance(sequence);
                    x.chain.is_from_BWAseed.assign(x.chain.size(), 0);
                    forReturn.extendWith(x.chain, false);
                    forReturn.dp_score[1] = x.score;
                }
            }
        }
        forReturn.checkConcordance(sequence);
        forReturn.extendToFull(sequence);
        return forReturn;
    }

    /* extensionAligner::scoreOneAlignment, extensionAligner.cpp:52-182.  `read_seq`/`read_qual` are the
     * oneRead in ORIGINAL orientation (processBAM.cpp:3147-3166). */
    double scoreOneAlignment(const Chain& alignment, const std::string& read_seq, const std::string& read_qual, bool longReadMode) const
    {
        int indexIntoOriginalReadData = alignment.sequence_begin - 1;
        double rate_deletions = log(0.001), rate_insertions = log(0.001);
        if(longReadMode) { rate_deletions = log(0.075); rate_insertions = log(0.075); }
        double rate_match_mismatch = log(1 - exp(rate_deletions) - exp(rate_insertions));
        double combined_log_likelihood = 0;
        for(size_t cI = 0; cI < alignment.sequence_aligned.size(); cI++) {
            char sequenceCharacter = alignment.sequence_aligned[cI];
            char graphCharacter = alignment.graph_aligned[cI];
            if(sequenceCharacter != '_') {
                indexIntoOriginalReadData++;
                int idx = indexIntoOriginalReadData;
                if(alignment.reverse) idx = (int)read_seq.size() - idx - 1;
                ORC_CHECK(idx >= 0 && idx < (int)read_seq.size(), "score index");
                if(!longReadMode) {
                    char u = read_seq[idx];
                    if(alignment.reverse) u = complementChar(u);
                    ORC_CHECK(u == sequenceCharacter, "score: read character mismatch");
                }
                if(graphCharacter == '_') {
                    combined_log_likelihood += (rate_insertions + log(1.0 / 4.0));
                } else {
                    combined_log_likelihood += rate_match_mismatch;
                    double pCorrect = PhredToPCorrect((unsigned char)read_qual.at(idx));
                    if(pCorrect > 0.999) pCorrect = 0.999;
                    if(pCorrect == 0) pCorrect = 0.00001;
                    ORC_CHECK(pCorrect > 0 && pCorrect <= 1, "pCorrect range");
                    if(sequenceCharacter == graphCharacter) combined_log_likelihood += log(pCorrect);
                    else { double pIncorrect = 1 - pCorrect; pIncorrect *= (1.0 / 3.0); combined_log_likelihood += log(pIncorrect); }
                }
            } else {
                if(graphCharacter != '_') combined_log_likelihood += rate_deletions;
            }
        }
        return combined_log_likelihood;
    }
};

/* ------------------------------------------------------------------------ processBAM */

struct Contigs {
    int n = 0;
    std::vector<long long> off;
    std::vector<unsigned char> seq;
    std::vector<int> level;
    std::vector<int> seqid;
    /* processBAM::graphLevel_2_underlyingSequencePositions, filled by _loadMapping (processBAM.cpp:4441-4456) */
    std::vector<std::map<int, int>> level2pos;
    void build(const hlala_contigs_desc* d, int L)
    {
        n = d->n_contigs;
        off.assign(d->contig_off, d->contig_off + n + 1);
        seq.assign(d->contig_seq, d->contig_seq + off[n]);
        level.assign(d->contig_level, d->contig_level + off[n]);
        seqid.assign(d->contig_seqid, d->contig_seqid + n);
        level2pos.assign(L, {});
        for(int c = 0; c < n; c++)
            for(long long p = off[c]; p < off[c + 1]; p++) {
                ORC_CHECK(level[p] >= 0 && level[p] < L, "translation level range");
                level2pos[level[p]][seqid[c]] = (int)(p - off[c]);
            }
    }
};

/* mapper::reads::PRGContigBAMAlignment, mapper/reads/PRGContigBAMAlignment.h */
struct ContigAlignment {
    std::vector<int> levels; std::string graph_aligned, sequence_aligned;
    int startInRaw = -1, stopInRaw = -1; bool reverse = false;
};

struct BamRecord { int contig, pos, offset, as; bool reverse; std::vector<uint32_t> cigar; };

static const char* CIGAR_OPS = "MIDNSHP=X";

struct Processor {
    Graph g;
    Contigs contigs;
    hlala_params params;
    Aligner* eA = nullptr;

    /* processBAM::transformBAMreadToInternalAlignment, mapper/processBAM.cpp:4794-5337 */
    bool transformBAMreadToInternalAlignment(const BamRecord& al, const std::string& queryBases, ContigAlignment& out) const
    {
        const unsigned char* referenceSequence = contigs.seq.data() + contigs.off[al.contig];
        const int* reference2level = contigs.level.data() + contigs.off[al.contig];
        long long refLen = contigs.off[al.contig + 1] - contigs.off[al.contig];
        out = ContigAlignment();
        out.reverse = al.reverse;
        std::vector<char> CIGAR;
        for(uint32_t c : al.cigar) {                                   /* :4814-4828, 'P' dropped */
            unsigned op = c & 15u, len = c >> 4;
            ORC_CHECK(op < 9, "cigar op");
            if(CIGAR_OPS[op] != 'P') CIGAR.insert(CIGAR.end(), len, CIGAR_OPS[op]);
        }
        ORC_CHECK(al.cigar.size() >= 1, "empty cigar");
        int index_along_genome_fromReadStart = 0, index_along_read = 0, index_along_unclipped_read = 0;
        if(CIGAR_OPS[al.cigar.front() & 15u] == 'H') index_along_unclipped_read += (int)(al.cigar.front() >> 4);     /* :4868-4874 */
        int readStart = al.pos;
        auto qb = [&](int i) -> char { ORC_CHECK(i >= 0 && i < (int)queryBases.size(), "read index in CIGAR walk"); return queryBases[i]; };
        auto ref = [&](int i) -> char { ORC_CHECK(i >= 0 && i < refLen, "reference index in CIGAR walk"); return (char)referenceSequence[i]; };
        for(size_t cigarI = 0; cigarI < CIGAR.size(); cigarI++) {
            int index_into_genome = readStart + index_along_genome_fromReadStart;
            std::string allele, genome; std::vector<int> genome_level;
            int allele_start = -1;
            char op = CIGAR[cigarI];
            switch(op) {
            case 'M': case '=': case 'X': case 'D':
                if(op == 'D') { allele = "_"; }
                else { allele = std::string(1, qb(index_along_unclipped_read)); }
                genome = std::string(1, ref(index_into_genome));
                genome_level.push_back(index_into_genome);
                allele_start = index_along_unclipped_read;
                while(cigarI + 1 < CIGAR.size() && CIGAR[cigarI + 1] == 'I') {               /* :5042-5065 */
                    int nextPositionUnclipped = (op == 'D') ? index_along_unclipped_read : index_along_unclipped_read + 1;
                    allele.push_back(qb(nextPositionUnclipped));
                    genome.push_back('_'); genome_level.push_back(-1);
                    index_along_read++; index_along_unclipped_read++; cigarI++;
                }
                index_along_genome_fromReadStart++; index_along_read++;
                if(op != 'D') index_along_unclipped_read++;
                break;
            case 'I': {                                                                      /* :5085-5166 */
                ORC_CHECK(index_along_read == 0, "internal I not attached to a column");
                std::string a1(1, qb(index_along_unclipped_read)); std::string g1 = "_"; std::vector<int> l1{-1};
                ORC_CHECK(out.startInRaw == -1, "leading I after start");
                out.startInRaw = index_along_unclipped_read;
                while(cigarI + 1 < CIGAR.size() && CIGAR[cigarI + 1] == 'I') {
                    a1.push_back(qb(index_along_unclipped_read + 1)); g1.push_back('_'); l1.push_back(-1);
                    index_along_read++; index_along_unclipped_read++; cigarI++;
                }
                index_along_read++; index_along_unclipped_read++;
                out.sequence_aligned.append(a1); out.graph_aligned.append(g1); out.levels.insert(out.levels.end(), l1.begin(), l1.end());
                out.stopInRaw = index_along_unclipped_read - 1;
                break; }
            case 'N': throw oracle_error("N character in CIGAR");
            case 'S': index_along_unclipped_read++; break;
            case 'H': break;
            default: throw oracle_error("Unknown element of CIGAR string");
            }
            if(allele != "") {                                                               /* :5185-5213 */
                out.sequence_aligned.append(allele); out.graph_aligned.append(genome);
                out.levels.insert(out.levels.end(), genome_level.begin(), genome_level.end());
                if(out.startInRaw == -1) out.startInRaw = allele_start;
                out.stopInRaw = index_along_unclipped_read - 1;
            }
        }
        ORC_CHECK(out.startInRaw < out.stopInRaw, "sequence_aligned_startInRaw < stopInRaw");   /* :5252 */
        bool haveNonMinusOne = false;
        for(int l : out.levels) if(l != -1) { haveNonMinusOne = true; break; }
        if(!haveNonMinusOne) return false;
        size_t positions_nonGap = 0;
        for(size_t c = 0; c < out.levels.size(); c++) {                                      /* :5290-5317 */
            if(out.levels[c] != -1) {
                int l = out.levels[c] - al.offset;
                ORC_CHECK(l >= 0 && l < refLen, "translation index");
                out.levels[c] = reference2level[l];
            }
            if(out.sequence_aligned[c] != '_') positions_nonGap++;
        }
        ORC_CHECK((long long)positions_nonGap == (out.stopInRaw - out.startInRaw + 1), "positions_nonGap");
        return true;
    }

    /* processBAM::cleanInitialAlignment, mapper/processBAM.cpp:4621-4792 */
    static void cleanInitialAlignment(std::vector<int>& lv, std::string& ga, std::string& sa)
    {
        bool inStretch = false; int stretchStart = -1; int balance = 0; bool cleaningNecessary = false;
        for(size_t pI = 0; pI < lv.size(); pI++) {
            if(lv[pI] == -1 || (ga[pI] == '_' && sa[pI] == '_')) {
                if(!inStretch) { stretchStart = (int)pI; inStretch = true; }
                if(lv[pI] == -1) balance++;
                if(ga[pI] == '_' && sa[pI] == '_') { ORC_CHECK(lv[pI] != -1, "double gap without level"); balance--; }
            } else if(inStretch) {
                int stretchStop = (int)pI - 1;
                if(balance == 0) {
                    int Ls = stretchStop - stretchStart + 1;
                    std::string inserted; std::vector<int> gapLevels;
                    for(int p = stretchStart; p <= stretchStop; p++) {
                        if(lv[p] == -1) inserted.push_back(sa[p]); else gapLevels.push_back(lv[p]);
                    }
                    ORC_CHECK(inserted.size() == gapLevels.size() && (int)inserted.size() == Ls / 2, "clean stretch balance");
                    cleaningNecessary = true;
                    for(int p = stretchStart; p <= stretchStop; p++) {
                        int i = p - stretchStart;
                        if(i < Ls / 2) { lv[p] = gapLevels[i]; ga[p] = '_'; sa[p] = inserted[i]; }
                        else { lv[p] = -1; ga[p] = '_'; sa[p] = '_'; }
                    }
                }
                inStretch = false; stretchStart = -1; balance = 0;
            }
        }
        if(cleaningNecessary) {
            std::vector<int> nl; std::string ng, ns;
            for(size_t pI = 0; pI < lv.size(); pI++)
                if(!(lv[pI] == -1 && ga[pI] == '_' && sa[pI] == '_')) { nl.push_back(lv[pI]); ng.push_back(ga[pI]); ns.push_back(sa[pI]); }
            lv = nl; ga = ng; sa = ns;
        }
    }

    /* processBAM::restrictInitialAlignmentToNoGapAreas, mapper/processBAM.cpp:4461-4619 */
    void restrictInitialAlignmentToNoGapAreas(std::vector<int>& lv, std::string& ga, std::string& sa, int& startInRaw, int& stopInRaw) const
    {
        int runningBegin = -1; int sequenceCharacters = 0;
        std::vector<std::pair<int, int>> pre;
        for(size_t lI = 0; lI < lv.size(); lI++) {
            if(sa[lI] != '_') sequenceCharacters++;
            int graph_level = lv[lI];
            if(graph_level != -1 && g.inGraphGapStretch.at(graph_level)) {
                if(runningBegin != -1) { pre.push_back({runningBegin, (int)lI - 1}); runningBegin = -1; }
            } else if(runningBegin == -1) runningBegin = (int)lI;
        }
        if(runningBegin != -1 && runningBegin != 0) {                               /* :4492-4502: a stretch from column 0 to the end is not recorded */
            int e = (int)lv.size() - 1;
            if(e >= runningBegin) pre.push_back({runningBegin, e});
        }
        std::vector<std::pair<int, int>> possible;
        for(auto s : pre) {
            while(lv.at(s.first) == -1) { s.first++; if(s.first > s.second || s.first > (int)lv.size() - 1) break; }
            while(lv.at(s.second) == -1) { s.second--; if(s.second < s.first || s.second < 0) break; }
            if(s.second >= s.first) possible.push_back(s);
        }
        /* std::sort by length ascending then .back() (:4533-4552): libstdc++ uses insertion sort below 16
         * elements, which is stable, so among equal longest stretches the LAST one in column order is taken. */
        std::stable_sort(possible.begin(), possible.end(), [](const std::pair<int, int>& a, const std::pair<int, int>& b) {
            return (a.second - a.first + 1) < (b.second - b.first + 1); });
        if(possible.size() > 0) {
            std::pair<int, int> sel = possible.back();
            int newStart = startInRaw, newStop = stopInRaw;
            for(int lI = 0; lI < sel.first; lI++) if(sa[lI] != '_') newStart++;
            for(int lI = sel.second + 1; lI < (int)lv.size(); lI++) if(sa[lI] != '_') newStop--;
            int stretchChars = 0;
            std::vector<int> nl; std::string ng, ns;
            for(int lI = sel.first; lI <= sel.second; lI++) { nl.push_back(lv[lI]); ng.push_back(ga[lI]); ns.push_back(sa[lI]); if(sa[lI] != '_') stretchChars++; }
            if(((double)stretchChars / (double)sequenceCharacters) > 0.3) { lv = nl; ga = ng; sa = ns; startInRaw = newStart; stopInRaw = newStop; }
        }
    }

    /* processBAM::PRGContigAlignment2Seed, mapper/processBAM.cpp:2491-3017; returns the sequenceSeed
     * (the only one alignment2Chain hands on, :3126).  `restrictGaps` = false reproduces the all-false
     * inGraphGapStretch vector of --action testChainExtension (HLA-LA.cpp:1812-1814). */
    Chain PRGContigAlignment2Seed(const ContigAlignment& al, bool restrictGaps) const
    {
        std::vector<int> lv; std::string ga, sa;
        int startInRaw = al.startInRaw, stopInRaw = al.stopInRaw;
        ORC_CHECK(al.levels.size() > 0, "empty contig alignment");
        size_t firstColumn = 0;
        while(al.levels.at(firstColumn) == -1) { firstColumn++; startInRaw++; }
        size_t lastColumn = al.levels.size() - 1;
        while(al.levels.at(lastColumn) == -1) { lastColumn--; stopInRaw--; }
        ORC_CHECK(firstColumn < lastColumn, "firstColumn < lastColumn");                      /* :2535 */
        int lastInserted = -1;
        for(size_t c = firstColumn; c <= lastColumn; c++) {                                    /* :2540-2579 */
            if(c == firstColumn) { lv.push_back(al.levels[c]); ga.push_back(al.graph_aligned[c]); sa.push_back(al.sequence_aligned[c]); lastInserted = al.levels[c]; }
            else if(al.levels[c] == -1) { lv.push_back(-1); ga.push_back(al.graph_aligned[c]); sa.push_back(al.sequence_aligned[c]); }
            else {
                if(lastInserted + 1 != al.levels[c]) {
                    ORC_CHECK(lastInserted + 1 < al.levels[c], "levels must increase");
                    for(int l = lastInserted + 1; l <= al.levels[c] - 1; l++) { lv.push_back(l); ga.push_back('_'); sa.push_back('_'); }
                }
                lv.push_back(al.levels[c]); ga.push_back(al.graph_aligned[c]); sa.push_back(al.sequence_aligned[c]); lastInserted = al.levels[c];
            }
        }
        cleanInitialAlignment(lv, ga, sa);
        int before = (int)lv.size();
        if(restrictGaps) restrictInitialAlignmentToNoGapAreas(lv, ga, sa, startInRaw, stopInRaw);
        int removed = before - (int)lv.size();
        size_t before_matches = 0;
        for(size_t i = 0; i < ga.size(); i++) if(ga[i] == sa[i]) before_matches++;
        ORC_CHECK(lv.front() != -1 && lv.back() != -1, "seed ends defined");

        /* sequence-variant path finding, :2676-2835 */
        struct SB { double S; std::set<int> takenEdges; };
        std::vector<std::map<int, SB>> bt(lv.size() + 1);
        ORC_CHECK(lv[0] < g.L, "first level range");
        for(int n : g.level_nodes.at(lv[0])) bt[0][n].S = 0;
        size_t lastNonGap = 0;
        for(size_t columnI = 1; columnI <= lv.size(); columnI++) {
            if(lv[columnI - 1] == -1) continue;
            int graphLevel = lv[columnI - 1];
            char sequenceCharacter = sa[columnI - 1], graphCharacter = ga[columnI - 1];
            bool seedIsMatch = (sequenceCharacter == graphCharacter);
            for(auto& nodeIt : bt[lastNonGap]) {
                int fromN = nodeIt.first;
                ORC_CHECK(g.node_level[fromN] == graphLevel, "rethreading level");
                for(int e : g.out_e[fromN]) {
                    t_stats.edges++;
                    if(seedIsMatch && (char)g.elabel[e] != sequenceCharacter) continue;
                    double S = ((char)g.elabel[e] == sequenceCharacter) ? 1 : 0;
                    int toN = g.eto[e];
                    double cand = nodeIt.second.S + S;
                    auto it = bt[columnI].find(toN);
                    if(it == bt[columnI].end()) { bt[columnI][toN].S = cand; bt[columnI][toN].takenEdges.insert(e); }
                    else if(it->second.S == cand) it->second.takenEdges.insert(e);
                    else if(it->second.S < cand) { it->second.takenEdges.clear(); it->second.takenEdges.insert(e); it->second.S = cand; }
                }
            }
            ORC_CHECK(bt[columnI].size() > 0, "rethreading found no edge");
            lastNonGap = columnI;
        }
        /* backtrace, :2838-2970 */
        std::vector<int> bl, be; std::string bg, bs;
        double Smax = 0; bool first = true;
        for(auto& n : bt[lv.size()]) if(first || n.second.S > Smax) { Smax = n.second.S; first = false; }
        int running = -1;
        for(auto& n : bt[lv.size()]) if(n.second.S == Smax) { running = n.first; break; }       /* *(runningN.begin()), :2867 */
        for(int columnI = (int)lv.size(); columnI >= 1; columnI--) {
            if(lv[columnI - 1] == -1) { bl.push_back(-1); be.push_back(-1); bg.push_back('_'); bs.push_back(sa[columnI - 1]); continue; }
            const SB& sb = bt[columnI].at(running);
            int e = *sb.takenEdges.begin();                                                       /* first edge in std::set<Edge*> order, :2906-2915 */
            bl.push_back(lv[columnI - 1]); be.push_back(e); bg.push_back((char)g.elabel[e]); bs.push_back(sa[columnI - 1]);
            running = g.efrom[e];
        }
        std::reverse(bl.begin(), bl.end()); std::reverse(be.begin(), be.end()); std::reverse(bg.begin(), bg.end()); std::reverse(bs.begin(), bs.end());
        Chain r;
        r.edges = be; r.levels = bl; r.graph_aligned = bg; r.sequence_aligned = bs;
        r.sequence_begin = startInRaw; r.sequence_end = stopInRaw; r.reverse = al.reverse;
        r.removed_columns_noGap_restriction = removed;
        size_t after_matches = 0;
        for(size_t i = 0; i < bg.size(); i++) if(bg[i] == bs[i]) after_matches++;
        r.improvement_through_bt = (double)after_matches / (double)bg.size() - (double)before_matches / (double)ga.size();
        r.is_from_BWAseed.assign(r.size(), 1);                                                    /* :3123-3124 */
        return r;
    }

    /* processBAM::alignment_get_startstop_PRGcoordinates, mapper/processBAM.cpp:3840-3898; GetEndPosition(false,true)
     * of BamTools 2.5.1 (un-vendored): Position + sum of M,=,X,D,N lengths - 1. */
    std::pair<int, int> startstop(const BamRecord& al) const
    {
        const int* tr = contigs.level.data() + contigs.off[al.contig];
        long long refLen = contigs.off[al.contig + 1] - contigs.off[al.contig];
        int readStart = al.pos, readStop = al.pos;
        for(uint32_t c : al.cigar) { char op = CIGAR_OPS[c & 15u]; if(op == 'M' || op == '=' || op == 'X' || op == 'D' || op == 'N') readStop += (int)(c >> 4); }
        readStop -= 1;
        int a = readStart - al.offset, b = readStop - al.offset;
        ORC_CHECK(a >= 0 && a < refLen && b >= 0 && b < refLen, "start/stop translation index");
        return {tr[a], tr[b]};
    }

    /* alignerBase::alignedReadPair_strandsValid, alignerBase.cpp:213-244 */
    static bool strandsValid(const Chain& a1, const Chain& a2)
    {
        if(a1.firstLevel() != -1 && a2.firstLevel() != -1 && a1.reverse != a2.reverse) {
            if(!a1.reverse) return a1.firstLevel() < a2.firstLevel();
            return a1.lastLevel() > a2.lastLevel();
        }
        return false;
    }
    /* verboseSeedChain::alignment_{end,begin}_originalSequenceAnchors, verboseSeedChain.h:230-280 */
    std::map<int, int> anchors(const std::vector<int>& levels_for_anchors) const
    {
        std::map<int, int> r;
        for(int level : levels_for_anchors)
            for(auto& it : contigs.level2pos.at(level))
                if(r.count(it.first) == 0) r[it.first] = it.second;
        return r;
    }
    /* alignerBase::alignedReadPair_pairsDistancesUnderlyingSequences, alignerBase.cpp:290-329 */
    std::set<int> pairDistances(const Chain& a1, const Chain& a2) const
    {
        std::set<int> r; const int scanPositions = 2;
        const Chain& up = (a1.firstLevel() < a2.firstLevel()) ? a1 : a2;
        const Chain& down = (a1.firstLevel() < a2.firstLevel()) ? a2 : a1;
        std::map<int, int> endA = anchors(up.lastLevels(scanPositions)), beginA = anchors(down.firstLevels(scanPositions));
        for(auto& it : endA) if(beginA.count(it.first)) r.insert(beginA.at(it.first) - it.second - 1);
        return r;
    }

    struct PairResult {
        int status = 0; int best1 = -1, best2 = -1; int nComb = 0; double ll = 0, mapQ = 0; bool strandsOK = false;
        Chain c1, c2;
    };

    /* processBAM::assignMappingQualities, mapper/processBAM.cpp:4062-4312 */
    void assignMappingQualities(PairResult& R, const std::vector<std::pair<unsigned, unsigned>>& idx, const std::vector<double>& LL,
                                std::pair<double, unsigned> mx, const std::vector<Chain>& r1, const std::vector<Chain>& r2) const
    {
        if(idx.size() > 1) {
            unsigned i1m = idx.at(mx.second).first, i2m = idx.at(mx.second).second;
            std::vector<double> PP = LL;
            for(double& p : PP) { p = exp(p - mx.first); ORC_CHECK(p >= 0 && p <= 1, "PP range"); }
            double S = 0; for(double p : PP) S += p;                                  /* Utilities::normalize_vector, Utilities.cpp:987-999 */
            for(double& p : PP) p = p / S;
            R.c1 = r1.at(i1m); R.c2 = r2.at(i2m);
            double mapQ = PP.at(mx.second);
            R.mapQ = mapQ;
            double q1 = 0, q2 = 0;
            for(size_t i = 0; i < PP.size(); i++) { if(idx[i].first == i1m) q1 += PP[i]; if(idx[i].second == i2m) q2 += PP[i]; }
            if(q1 > 1) q1 = 1; if(q2 > 1) q2 = 1;
            R.c1.mapQ = q1; R.c2.mapQ = q2;
            /* key "c:level:rN:strand:idx" (:4175, 4190) as a tuple -- same key set, same accumulation order */
            typedef std::tuple<int, char, int, int, int> PKey;    /* mate, graph char, level, strand, sequence index */
            std::map<PKey, double> conf;
            auto seqIndex = [](const std::string& sa, bool reverse) {               /* alignedSequence2SequenceIndex, :4117-4153 */
                std::vector<int> r; int noGap = 0; for(char c : sa) if(c != '_') noGap++;
                int i_noGap = -1;
                for(char c : sa) { if(c == '_') r.push_back(-1); else { i_noGap++; r.push_back(reverse ? noGap - i_noGap - 1 : i_noGap); } }
                return r;
            };
            for(size_t i = 0; i < idx.size(); i++) {
                const Chain& a = r1.at(idx[i].first); const Chain& b = r2.at(idx[i].second);
                std::vector<int> ia = seqIndex(a.sequence_aligned, a.reverse), ib = seqIndex(b.sequence_aligned, b.reverse);
                for(size_t j = 0; j < a.size(); j++) conf[PKey(1, a.graph_aligned[j], a.levels[j], a.reverse, ia[j])] += PP[i];
                for(size_t j = 0; j < b.size(); j++) conf[PKey(2, b.graph_aligned[j], b.levels[j], b.reverse, ib[j])] += PP[i];
            }
            auto perPos = [&](Chain& c, int mate) {
                std::vector<int> ic = seqIndex(c.sequence_aligned, c.reverse);
                c.mapQ_perPosition.clear();
                for(size_t j = 0; j < c.size(); j++) {
                    double Q = conf.at(PKey(mate, c.graph_aligned[j], c.levels[j], c.reverse, ic[j]));
                    ORC_CHECK((Q - 1) <= 1e-5, "position confidence > 1");
                    if(Q > 1) Q = 1;
                    c.mapQ_perPosition.push_back((char)PCorrectToPhred(Q));
                }
            };
            perPos(R.c1, 1); perPos(R.c2, 2);
        } else {
            R.mapQ = 1; R.c1.mapQ = 1; R.c2.mapQ = 1;
            char Phred1 = (char)PCorrectToPhred(1);
            R.c1.mapQ_perPosition.assign(R.c1.size(), Phred1);
            R.c2.mapQ_perPosition.assign(R.c2.size(), Phred1);
        }
    }
};

static std::string invertRead(const std::string& s, bool complement)
{
    std::string r(s.rbegin(), s.rend());
    if(complement) for(char& c : r) c = complementChar(c);
    return r;
}

} /* namespace orc */

/* ============================================================================ C interface */

using namespace orc;

struct orc_handle { Processor P; std::string err; };

static thread_local std::string g_err;

static void storeChain(const Chain& c, int ci, int stride, hlala_chains_out* o, int status)
{
    if(!o) return;
    if(o->status) o->status[ci] = status;
    if(status != HLALA_CHAIN_OK) { if(o->n_cols) o->n_cols[ci] = 0; return; }
    int n = (int)c.size();
    if(n > stride) { if(o->status) o->status[ci] = HLALA_CHAIN_ERR_COLUMNS; if(o->n_cols) o->n_cols[ci] = 0; return; }
    if(o->n_cols) o->n_cols[ci] = n;
    if(o->seq_begin) o->seq_begin[ci] = c.sequence_begin;
    if(o->seq_end) o->seq_end[ci] = c.sequence_end;
    if(o->removed_cols) o->removed_cols[ci] = c.removed_columns_noGap_restriction;
    if(o->ll) o->ll[ci] = c.ll;
    if(o->dp_iters) { o->dp_iters[2 * ci] = c.dp_iters[0]; o->dp_iters[2 * ci + 1] = c.dp_iters[1]; }
    if(o->dp_score) { o->dp_score[2 * ci] = c.dp_score[0]; o->dp_score[2 * ci + 1] = c.dp_score[1]; }
    size_t base = (size_t)ci * stride;
    for(int j = 0; j < n; j++) {
        if(o->col_level) o->col_level[base + j] = c.levels[j];
        if(o->col_edge) o->col_edge[base + j] = c.edges[j];
        if(o->col_gchar) o->col_gchar[base + j] = (uint8_t)c.graph_aligned[j];
        if(o->col_schar) o->col_schar[base + j] = (uint8_t)c.sequence_aligned[j];
        if(o->col_fromseed) o->col_fromseed[base + j] = c.is_from_BWAseed[j];
    }
}

extern "C" {

const char* orc_last_error() { return g_err.c_str(); }

orc_handle* orc_create(const hlala_graph_desc* graph, const hlala_contigs_desc* contigs, const hlala_params* params)
{
    try {
        orc_handle* h = new orc_handle();
        h->P.g.build(graph);
        if(contigs) h->P.contigs.build(contigs, graph->n_levels);
        h->P.params = *params;
        h->P.eA = new Aligner(&h->P.g);
        return h;
    } catch(std::exception& e) { g_err = e.what(); return nullptr; }
}
void orc_destroy(orc_handle* h) { if(h) { delete h->P.eA; delete h; } }

/* largest frontier / candidate-cell set / kept-cell table / sequence-complete set of any DP call so far; reset != 0 clears them */
int orc_dp_maxima(orc_handle* h, int64_t* out4, int reset)
{
    (void)h;
    out4[0] = t_stats.max_frontier; out4[1] = t_stats.max_targets; out4[2] = t_stats.max_kept_cells; out4[3] = t_stats.max_completed;
    if(reset) t_stats.max_frontier = t_stats.max_targets = t_stats.max_kept_cells = t_stats.max_completed = 0;
    return 0;
}

/* per DP call of this thread so far: how many calls had their widest frontier / target set in (2^(b-1), 2^b]; out32 = 16 + 16 counts */
int orc_dp_histogram(orc_handle* h, int64_t* out32, int reset)
{
    (void)h;
    for(int i = 0; i < 16; i++) { out32[i] = t_stats.hist_frontier[i]; out32[16 + i] = t_stats.hist_targets[i]; }
    out32[15] = t_stats.ovf_calls; out32[30] = t_stats.ovf_first_iter; out32[31] = t_stats.ovf_total_iter;   /* (the last buckets are never reached: reused) */
    if(reset) for(int i = 0; i < 16; i++) t_stats.hist_frontier[i] = t_stats.hist_targets[i] = 0;
    return 0;
}

int orc_graph_info(orc_handle* h, hlala_graph_info* info)
{
    const Graph& g = h->P.g;
    memset(info, 0, sizeof(*info));
    info->n_levels = g.L; info->n_nodes = (int)g.node_level.size(); info->n_edges = (int)g.efrom.size(); info->n_paths = (int)g.paths.size();
    for(auto& a : g.jump_fwd) info->n_jump_entries += (int64_t)a.second.size();
    for(auto& p : g.paths) info->n_path_edges += (int64_t)p.size();
    for(auto& m : h->P.contigs.level2pos) info->n_levelpos_entries += (int64_t)m.size();
    for(auto& l : g.level_nodes) info->max_nodes_per_level = std::max(info->max_nodes_per_level, (int)l.size());
    for(auto& l : g.out_e) info->max_out_degree = std::max(info->max_out_degree, (int)l.size());
    for(auto& l : g.in_e) info->max_in_degree = std::max(info->max_in_degree, (int)l.size());
    for(auto b : g.inGraphGapStretch) info->n_gap_stretch_levels += b;
    for(auto& a : g.jump_fwd) info->max_jumps = std::max(info->max_jumps, (int)a.second.size());
    for(auto& a : g.jump_bwd) info->max_jumps = std::max(info->max_jumps, (int)a.second.size());
    /* most parallel edges between one pair of nodes (a gap path is unique per node pair: "duplicate gap path" above) */
    for(auto& l : g.out_e) { std::map<int, int> cnt; for(int e : l) info->max_parallel = std::max(info->max_parallel, ++cnt[g.eto[e]]); }
    return 0;
}
int orc_graph_get_paths(orc_handle* h, int32_t* first_node, int32_t* last_node, int32_t* length)
{
    const Graph& g = h->P.g;
    for(size_t i = 0; i < g.paths.size(); i++) { first_node[i] = g.efrom[g.paths[i].front()]; last_node[i] = g.eto[g.paths[i].back()]; length[i] = (int)g.paths[i].size(); }
    return 0;
}
int orc_graph_get_gap_stretch(orc_handle* h, uint8_t* out)
{
    memcpy(out, h->P.g.inGraphGapStretch.data(), h->P.g.inGraphGapStretch.size());
    return 0;
}

/* known-answer helpers */
int orc_intervals_overlap(int x1, int x2, int y1, int y2) { return intervalsOverlap(x1, x2, y1, y2) ? 1 : 0; }
int orc_phred(int n, const double* p_correct, uint8_t* phred_out, const uint8_t* phred_in, double* p_out)
{
    for(int i = 0; i < n; i++) {
        if(p_correct && phred_out) phred_out[i] = PCorrectToPhred(p_correct[i]);
        if(phred_in && p_out) p_out[i] = PhredToPCorrect(phred_in[i]);
    }
    return 0;
}
int orc_rand_r(int n, uint32_t* seeds_inout, int32_t* values_out)
{
    for(int i = 0; i < n; i++) { unsigned int s = seeds_inout[i]; values_out[i] = rand_r(&s); seeds_inout[i] = s; }
    return 0;
}
double orc_normal_logpdf_penalty(double mean, double sd) { return log(normal_pdf(mean, sd, mean + 8 * sd)); }

/* extendSeedChain + scoreOneAlignment over seed chains handed in directly (testChainExtension protocol).
 * Read r's bases are in alignment orientation; scoring needs the original-orientation read
 * (processBAM.cpp:3147-3166), derived here from the chain's reverse flag. */
int orc_extend_seeds(orc_handle* h, const hlala_seeds_in* in, hlala_chains_out* out, int64_t* stats /* [4] calls, iters, cells, edges */)
{
    try {
        Processor& P = h->P;
        Aligner& A = *P.eA;
        t_stats.calls = t_stats.iters = t_stats.cells = t_stats.edges = 0;
        int stride = P.params.max_columns;
        for(int c = 0; c < in->n_chains; c++) {
            int r = in->chain_read[c];
            std::string seq((const char*)in->read_bases + in->read_off[r], in->read_off[r + 1] - in->read_off[r]);
            std::string qual((const char*)in->read_quals + in->read_off[r], in->read_off[r + 1] - in->read_off[r]);
            Chain s;
            s.sequence_begin = in->chain_seq_begin[c]; s.sequence_end = in->chain_seq_end[c]; s.reverse = in->chain_reverse[c] != 0;
            for(int j = in->col_off[c]; j < in->col_off[c + 1]; j++) {
                s.levels.push_back(in->col_level[j]); s.edges.push_back(in->col_edge[j]);
                s.graph_aligned.push_back((char)in->col_gchar[j]); s.sequence_aligned.push_back((char)in->col_schar[j]);
            }
            s.is_from_BWAseed.assign(s.size(), 1);
            Chain e = A.extendSeedChain(seq, s, P.params.rng_seed + 2u * (unsigned)c, P.params.rng_seed + 2u * (unsigned)c + 1u);
            e.checkLevelContiguity();
            std::string oseq = s.reverse ? invertRead(seq, true) : seq;
            std::string oqual = s.reverse ? invertRead(qual, false) : qual;
            e.ll = A.scoreOneAlignment(e, oseq, oqual, P.params.long_read_mode != 0);
            storeChain(e, c, stride, out, HLALA_CHAIN_OK);
        }
        if(stats) { stats[0] = t_stats.calls; stats[1] = t_stats.iters; stats[2] = t_stats.cells; stats[3] = t_stats.edges; }
        return 0;
    } catch(std::exception& e) { g_err = e.what(); return -1; }
}

/* processBAM::alignOneReadPair (mapper/processBAM.cpp:3129-3616) over a batch; optionally stops after the
 * projection stage.  seeds_out / ext_out / pairs_out may be NULL. */
/* one pair of processBAM::alignOneReadPair (mapper/processBAM.cpp:3129-3616); pairs are independent, the aligner is read-only */
static void align_one_pair(Processor& P, const hlala_batch_in* in, int p, hlala_chains_out* seeds_out, hlala_chains_out* ext_out,
                           hlala_pairs_out* pairs_out, int stop_after_projection)
{
    Aligner& A = *P.eA;
    const int stride = P.params.max_columns;
    const bool longRead = P.params.long_read_mode != 0;
    const double IS_mean = P.params.insert_mean, IS_sd = P.params.insert_sd;
    const double max_insertsize_penalty_log = log(normal_pdf(IS_mean, IS_sd, IS_mean + 8 * IS_sd));   /* processBAM.cpp:2342-2346 */
    {
        std::vector<Chain> ext[2]; std::vector<double> ll[2]; std::vector<int> extIdx[2];
        bool pairErr = false;
        for(int m = 0; m < 2; m++) {
            int r = 2 * p + m;
            std::string seq((const char*)in->read_bases + in->read_off[r], in->read_off[r + 1] - in->read_off[r]);
            std::string qual((const char*)in->read_quals + in->read_off[r], in->read_off[r + 1] - in->read_off[r]);
            int prim = in->read_primary[r];
            bool primReverse = in->chain_reverse[prim] != 0;
            /* oneRead in original orientation: invert() if the primary is on the reverse strand (:3147-3166) */
            std::string oseq = primReverse ? invertRead(seq, true) : seq;
            std::string oqual = primReverse ? invertRead(qual, false) : qual;
            std::map<std::string, int> alignments_scores;                                            /* :3198 */
            for(int c = in->chain_off[r]; c < in->chain_off[r + 1]; c++) {
                BamRecord al; al.contig = in->chain_contig[c]; al.pos = in->chain_pos[c]; al.offset = in->chain_offset[c];
                al.as = in->chain_as[c]; al.reverse = in->chain_reverse[c] != 0;
                al.cigar.assign(in->cigar + in->cigar_off[c], in->cigar + in->cigar_off[c + 1]);
                std::pair<int, int> ss = P.startstop(al);
                std::string id = std::to_string(ss.first) + "//" + std::to_string(ss.second);
                int score = al.as;
                if(al.reverse != primReverse) {                                                     /* :3216 */
                    storeChain(Chain(), c, stride, seeds_out, HLALA_CHAIN_SKIP_STRAND);
                    storeChain(Chain(), c, stride, ext_out, HLALA_CHAIN_SKIP_STRAND);
                    continue;
                }
                /* alignment2Chain (:3019-3127) is evaluated BEFORE the duplicate test in the reference (:3225 vs :3234);
                 * its result is discarded for duplicates, so only its asserts could matter. */
                ContigAlignment ca;
                bool ok = P.transformBAMreadToInternalAlignment(al, seq, ca);
                ORC_CHECK(ok, "alignment consists of insertions only");
                Chain seed = P.PRGContigAlignment2Seed(ca, true);
                if(alignments_scores.count(id) && alignments_scores.at(id) >= score) {               /* :3234 */
                    storeChain(Chain(), c, stride, seeds_out, HLALA_CHAIN_SKIP_DUP);
                    storeChain(Chain(), c, stride, ext_out, HLALA_CHAIN_SKIP_DUP);
                    continue;
                }
                seed.checkConcordance(seq);
                storeChain(seed, c, stride, seeds_out, HLALA_CHAIN_OK);
                if(stop_after_projection) { if(alignments_scores.count(id) == 0 || alignments_scores.at(id) < score) alignments_scores[id] = score; continue; }
                Chain e = A.extendSeedChain(seq, seed, P.params.rng_seed + 2u * (unsigned)c, P.params.rng_seed + 2u * (unsigned)c + 1u);
                e.ll = A.scoreOneAlignment(e, oseq, oqual, longRead);
                storeChain(e, c, stride, ext_out, HLALA_CHAIN_OK);
                if((int)e.size() > stride) pairErr = true;
                ext[m].push_back(e); ll[m].push_back(e.ll); extIdx[m].push_back(c);
                if(alignments_scores.count(id) == 0 || alignments_scores.at(id) < score) alignments_scores[id] = score;
            }
        }
        if(stop_after_projection || !pairs_out) return;
        ORC_CHECK(ext[0].size() > 0 && ext[1].size() > 0, "no extended chains for a mate");           /* :3393-3394 */
        /* pairing loop, :3408-3506 */
        std::vector<std::pair<unsigned, unsigned>> idx; std::vector<double> LL;
        for(unsigned i1 = 0; i1 < ext[0].size(); i1++)
            for(unsigned i2 = 0; i2 < ext[1].size(); i2++) {
                double combined = ll[0][i1] + ll[1][i2];
                const Chain& c1 = ext[0][i1]; const Chain& c2 = ext[1][i2];
                double ll_IS;
                if(Processor::strandsValid(c1, c2)) {
                    std::set<int> dist = P.pairDistances(c1, c2);
                    if(dist.size()) {
                        std::vector<double> lls;
                        for(int d : dist) {
                            double dP = normal_pdf(IS_mean, IS_sd, d);
                            if(dP <= 0) lls.push_back(max_insertsize_penalty_log); else lls.push_back(log(dP));
                        }
                        ll_IS = firstMax(lls).first;
                    } else ll_IS = max_insertsize_penalty_log;
                } else ll_IS = max_insertsize_penalty_log;
                combined += ll_IS;
                LL.push_back(combined); idx.push_back({i1, i2});
            }
        auto mx = firstMax(LL);                                                                    /* :3538 */
        Processor::PairResult R;
        R.best1 = idx[mx.second].first; R.best2 = idx[mx.second].second; R.nComb = (int)idx.size(); R.ll = mx.first;
        R.c1 = ext[0][R.best1]; R.c2 = ext[1][R.best2];
        P.assignMappingQualities(R, idx, LL, mx, ext[0], ext[1]);
        R.strandsOK = Processor::strandsValid(R.c1, R.c2);
        if(pairs_out->pair_status) pairs_out->pair_status[p] = pairErr ? -1 : 0;
        if(pairs_out->best_chain) { pairs_out->best_chain[2 * p] = extIdx[0][R.best1]; pairs_out->best_chain[2 * p + 1] = extIdx[1][R.best2]; }
        if(pairs_out->n_combinations) pairs_out->n_combinations[p] = R.nComb;
        if(pairs_out->pair_ll) pairs_out->pair_ll[p] = R.ll;
        if(pairs_out->pair_mapq) pairs_out->pair_mapq[p] = R.mapQ;
        if(pairs_out->mate_mapq) { pairs_out->mate_mapq[2 * p] = R.c1.mapQ; pairs_out->mate_mapq[2 * p + 1] = R.c2.mapQ; }
        if(pairs_out->strands_valid) pairs_out->strands_valid[p] = R.strandsOK ? 1 : 0;
        for(int m = 0; m < 2; m++) {
            const Chain& c = m ? R.c2 : R.c1;
            int r = 2 * p + m; int n = (int)c.size();
            if(n > stride) { if(pairs_out->n_cols) pairs_out->n_cols[r] = 0; continue; }
            if(pairs_out->n_cols) pairs_out->n_cols[r] = n;
            size_t base = (size_t)r * stride;
            for(int j = 0; j < n; j++) {
                if(pairs_out->col_level) pairs_out->col_level[base + j] = c.levels[j];
                if(pairs_out->col_edge) pairs_out->col_edge[base + j] = c.edges[j];
                if(pairs_out->col_gchar) pairs_out->col_gchar[base + j] = (uint8_t)c.graph_aligned[j];
                if(pairs_out->col_schar) pairs_out->col_schar[base + j] = (uint8_t)c.sequence_aligned[j];
                if(pairs_out->col_fromseed) pairs_out->col_fromseed[base + j] = c.is_from_BWAseed[j];
                if(pairs_out->col_mapq) pairs_out->col_mapq[base + j] = (uint8_t)c.mapQ_perPosition[j];
            }
        }
    }
}

int orc_align_batch(orc_handle* h, const hlala_batch_in* in, hlala_chains_out* seeds_out, hlala_chains_out* ext_out,
                    hlala_pairs_out* pairs_out, int stop_after_projection, int64_t* stats)
{
    try {
        Processor& P = h->P;
        t_stats.calls = t_stats.iters = t_stats.cells = t_stats.edges = 0;
        for(int p = 0; p < in->n_pairs; p++) align_one_pair(P, in, p, seeds_out, ext_out, pairs_out, stop_after_projection);
        if(stats) { stats[0] = t_stats.calls; stats[1] = t_stats.iters; stats[2] = t_stats.cells; stats[3] = t_stats.edges; }
        return 0;
    } catch(std::exception& e) { g_err = e.what(); return -1; }
}

/* The same over all host cores ("OpenMP parallel for schedule(dynamic,64) over pairs -- legal because pairs are
 * independent"; the reference itself runs this loop on one thread, HLA-LA.cpp:799).  n_threads <= 0: omp_get_max_threads().
 * Results are identical to orc_align_batch (every pair writes its own output rows; random seeds are per chain). */
int orc_align_batch_mt(orc_handle* h, const hlala_batch_in* in, hlala_chains_out* seeds_out, hlala_chains_out* ext_out,
                       hlala_pairs_out* pairs_out, int n_threads, int64_t* stats, int* threads_used)
{
    Processor& P = h->P;
    if(n_threads <= 0) n_threads = omp_get_max_threads();
    DpStats total; std::string firstErr; int used = 1;
#pragma omp parallel num_threads(n_threads)
    {
        t_stats = DpStats();
#pragma omp single
        used = omp_get_num_threads();
#pragma omp for schedule(dynamic, 64)
        for(int p = 0; p < in->n_pairs; p++) {
            try { align_one_pair(P, in, p, seeds_out, ext_out, pairs_out, 0); }
            catch(std::exception& e) {
#pragma omp critical
                if(firstErr.empty()) firstErr = e.what();
            }
        }
#pragma omp critical
        total.add(t_stats);
    }
    if(threads_used) *threads_used = used;
    if(stats) { stats[0] = total.calls; stats[1] = total.iters; stats[2] = total.cells; stats[3] = total.edges; }
    if(!firstErr.empty()) { g_err = firstErr; return -1; }
    return 0;
}

/* processBAM::alignOneLongRead (mapper/processBAM.cpp:3618-3838) + assignMappingQualities_unpaired (:3900-4059) for a batch of single
 * reads: in->n_pairs READS, per-read arrays have one entry per read.  pairs_out is filled per read ([n] where the paired layout has [2n]). */
int orc_align_long_reads(orc_handle* h, const hlala_batch_in* in, hlala_chains_out* seeds_out, hlala_chains_out* ext_out, hlala_pairs_out* pairs_out)
{
    try {
        Processor& P = h->P;
        Aligner& A = *P.eA;
        int stride = P.params.max_columns;
        bool longRead = P.params.long_read_mode != 0;
        for(int r = 0; r < in->n_pairs; r++) {
            std::string seq((const char*)in->read_bases + in->read_off[r], in->read_off[r + 1] - in->read_off[r]);
            std::string qual((const char*)in->read_quals + in->read_off[r], in->read_off[r + 1] - in->read_off[r]);
            int prim = in->read_primary[r];
            bool primReverse = in->chain_reverse[prim] != 0;
            std::string oseq = primReverse ? invertRead(seq, true) : seq;                        /* r1.invert(), :3636-3639 */
            std::string oqual = primReverse ? invertRead(qual, false) : qual;
            std::vector<Chain> read1_extendedChains; std::vector<double> read1_extendedChains_log_likelihoods; std::vector<int> extIdx;
            std::map<std::string, int> read1_alignments_scores;
            bool err = false;
            for(int c = in->chain_off[r]; c < in->chain_off[r + 1]; c++) {
                BamRecord al; al.contig = in->chain_contig[c]; al.pos = in->chain_pos[c]; al.offset = in->chain_offset[c];
                al.as = in->chain_as[c]; al.reverse = in->chain_reverse[c] != 0;
                al.cigar.assign(in->cigar + in->cigar_off[c], in->cigar + in->cigar_off[c + 1]);
                std::pair<int, int> ss = P.startstop(al);
                std::string id = std::to_string(ss.first) + "//" + std::to_string(ss.second);
                int score = al.as;
                if(al.reverse != primReverse) {                                                     /* :3685 */
                    storeChain(Chain(), c, stride, seeds_out, HLALA_CHAIN_SKIP_STRAND); storeChain(Chain(), c, stride, ext_out, HLALA_CHAIN_SKIP_STRAND);
                    continue;
                }
                ContigAlignment ca;
                bool ok = P.transformBAMreadToInternalAlignment(al, seq, ca);
                ORC_CHECK(ok, "alignment consists of insertions only");
                Chain al_Chain = P.PRGContigAlignment2Seed(ca, true);                                /* alignment2Chain, :3694 */
                if(read1_alignments_scores.count(id) && (read1_alignments_scores.at(id) >= score)) {  /* :3705 */
                    storeChain(Chain(), c, stride, seeds_out, HLALA_CHAIN_SKIP_DUP); storeChain(Chain(), c, stride, ext_out, HLALA_CHAIN_SKIP_DUP);
                    continue;
                }
                al_Chain.checkConcordance(seq);
                storeChain(al_Chain, c, stride, seeds_out, HLALA_CHAIN_OK);
                Chain al_Chain_extended = al_Chain;
                al_Chain_extended.extendToFull(seq);                                                  /* :3734-3735, no extension DP */
                al_Chain_extended.checkConcordance(seq);
                al_Chain_extended.ll = A.scoreOneAlignment(al_Chain_extended, oseq, oqual, longRead);
                storeChain(al_Chain_extended, c, stride, ext_out, HLALA_CHAIN_OK);
                if((int)al_Chain_extended.size() > stride) err = true;
                read1_extendedChains.push_back(al_Chain_extended); read1_extendedChains_log_likelihoods.push_back(al_Chain_extended.ll); extIdx.push_back(c);
                if((read1_alignments_scores.count(id) == 0) || (read1_alignments_scores.at(id) < score)) read1_alignments_scores[id] = score;
            }
            if(!pairs_out) continue;
            ORC_CHECK(read1_extendedChains.size() > 0, "no chains for a read");                      /* :3768 */
            auto mx = firstMax(read1_extendedChains_log_likelihoods);                               /* findVectorMax, :3770 */
            /* assignMappingQualities_unpaired == the paired routine with one neutral second mate */
            std::vector<std::pair<unsigned, unsigned>> idx; for(unsigned i = 0; i < read1_extendedChains.size(); i++) idx.push_back({i, 0u});
            std::vector<Chain> dummy(1);
            Processor::PairResult R; R.best1 = mx.second; R.best2 = 0; R.nComb = (int)idx.size(); R.ll = mx.first;
            R.c1 = read1_extendedChains[mx.second];
            P.assignMappingQualities(R, idx, read1_extendedChains_log_likelihoods, mx, read1_extendedChains, dummy);
            if(pairs_out->pair_status) pairs_out->pair_status[r] = err ? -1 : 0;
            if(pairs_out->best_chain) pairs_out->best_chain[r] = extIdx[R.best1];
            if(pairs_out->n_combinations) pairs_out->n_combinations[r] = R.nComb;
            if(pairs_out->pair_ll) pairs_out->pair_ll[r] = R.ll;
            if(pairs_out->pair_mapq) pairs_out->pair_mapq[r] = R.mapQ;
            if(pairs_out->mate_mapq) pairs_out->mate_mapq[r] = R.mapQ;                               /* forReturn.mapQ = mapQ, :3921 */
            if(pairs_out->strands_valid) pairs_out->strands_valid[r] = 0;
            const Chain& cc = R.c1; int n = (int)cc.size();
            if(n > stride) { if(pairs_out->n_cols) pairs_out->n_cols[r] = 0; continue; }
            if(pairs_out->n_cols) pairs_out->n_cols[r] = n;
            size_t base = (size_t)r * stride;
            for(int j = 0; j < n; j++) {
                if(pairs_out->col_level) pairs_out->col_level[base + j] = cc.levels[j];
                if(pairs_out->col_edge) pairs_out->col_edge[base + j] = cc.edges[j];
                if(pairs_out->col_gchar) pairs_out->col_gchar[base + j] = (uint8_t)cc.graph_aligned[j];
                if(pairs_out->col_schar) pairs_out->col_schar[base + j] = (uint8_t)cc.sequence_aligned[j];
                if(pairs_out->col_fromseed) pairs_out->col_fromseed[base + j] = cc.is_from_BWAseed[j];
                if(pairs_out->col_mapq) pairs_out->col_mapq[base + j] = (uint8_t)cc.mapQ_perPosition[j];
            }
        }
        return 0;
    } catch(std::exception& e) { g_err = e.what(); return -1; }
}

/* processBAM::calculateInsertSizeFromHistogram, mapper/processBAM.cpp:991-1069 */
static int calculateInsertSizeFromHistogram(const std::map<int, double>& IS_combined_counts, double& mean, double& sd, double& total)
{
    std::set<int> IS_keys; double IS_total_size = 0;
    for(auto ISentry : IS_combined_counts) { IS_keys.insert(ISentry.first); if(!(ISentry.second >= 0)) return -1; IS_total_size += ISentry.second; }
    double cumulative_sum = 0, weighted_median = 0, weighted_20 = 0, weighted_80 = 0;
    bool set_median = false, set_weighted_20 = false, set_weighted_80 = false;
    for(std::set<int>::iterator ISit = IS_keys.begin(); ISit != IS_keys.end(); ISit++) {
        int d = *ISit;
        cumulative_sum += IS_combined_counts.at(d);
        if((set_median == false) && (cumulative_sum >= (IS_total_size * 0.5))) { weighted_median = d; set_median = true; }
        if((set_weighted_20 == false) && (cumulative_sum >= (IS_total_size * 0.2))) { weighted_20 = d; set_weighted_20 = true; }
        if((set_weighted_80 == false) && (cumulative_sum >= (IS_total_size * 0.8))) { weighted_80 = d; set_weighted_80 = true; }
    }
    if(!(set_weighted_80 && set_weighted_20 && set_median)) return -2;                  // assert, :1044
    double f_mean_ret = weighted_median;
    weighted_20 = std::abs(weighted_median - weighted_20);                               // std::abs(double): Utilities.h:20 brings std into scope
    weighted_80 = std::abs(weighted_median - weighted_80);
    double f_sd_ret = (weighted_20 > weighted_80) ? weighted_20 : weighted_80;
    mean = f_mean_ret; sd = f_sd_ret; total = IS_total_size;
    return 0;
}

int orc_insert_size_from_histogram(int n, const int32_t* keys, const double* counts, double* mean, double* sd)
{
    std::map<int, double> h; for(int i = 0; i < n; i++) h[keys[i]] = counts[i];
    double total; return calculateInsertSizeFromHistogram(h, *mean, *sd, total);
}

/* processBAM::estimateInsertSize, mapper/processBAM.cpp:1071-1165, on the pairs of a batch in their given order: the primary alignment of
 * each mate (the first primary x primary combination, :1102-1153), alignment2Chain, extendSeedChain, strands, distances, histogram. */
int orc_estimate_insert_size(orc_handle* h, const hlala_batch_in* in, hlala_insert_size_out* out)
{
    try {
        Processor& P = h->P;
        Aligner& A = *P.eA;
        std::map<int, double> IS_combined_counts;
        int used_proto_seeds = 0, skipped_proto_seeds = 0;
        for(int p = 0; p < in->n_pairs; p++) {
            Chain ext[2];
            for(int m = 0; m < 2; m++) {
                int r = 2 * p + m;
                std::string seq((const char*)in->read_bases + in->read_off[r], in->read_off[r + 1] - in->read_off[r]);
                int c = in->read_primary[r];
                BamRecord al; al.contig = in->chain_contig[c]; al.pos = in->chain_pos[c]; al.offset = in->chain_offset[c];
                al.as = in->chain_as[c]; al.reverse = in->chain_reverse[c] != 0;
                al.cigar.assign(in->cigar + in->cigar_off[c], in->cigar + in->cigar_off[c + 1]);
                ContigAlignment ca;
                bool ok = P.transformBAMreadToInternalAlignment(al, seq, ca);
                ORC_CHECK(ok, "alignment consists of insertions only");
                Chain seed = P.PRGContigAlignment2Seed(ca, true);
                seed.checkConcordance(seq);
                // imposed determinism: the DP of mate r draws the seed rng_seed + 2r + side, as if the batch held the primaries only
                ext[m] = A.extendSeedChain(seq, seed, P.params.rng_seed + 2u * (unsigned)r, P.params.rng_seed + 2u * (unsigned)r + 1u);
            }
            used_proto_seeds++;
            if(Processor::strandsValid(ext[0], ext[1])) {
                std::set<int> underlyingSequencesDistances = P.pairDistances(ext[0], ext[1]);
                for(std::set<int>::iterator ISiterator = underlyingSequencesDistances.begin(); ISiterator != underlyingSequencesDistances.end(); ISiterator++) {
                    int IS = *ISiterator;
                    if(IS_combined_counts.count(IS) == 0) IS_combined_counts[IS] = 0;
                    double weight = 1.0 / (double)underlyingSequencesDistances.size();
                    IS_combined_counts[IS] += weight;
                }
            } else skipped_proto_seeds++;
        }
        out->n_used = used_proto_seeds; out->n_skipped = skipped_proto_seeds;
        int rc = calculateInsertSizeFromHistogram(IS_combined_counts, out->mean, out->sd, out->total_weight);
        if(rc) { g_err = "insert-size histogram is empty"; return rc; }
        return 0;
    } catch(std::exception& e) { g_err = e.what(); return -1; }
}

/* HLATyper per-cluster x per-read log-likelihood and mismatch count, hla/HLATyper.cpp:2067-2277 (parameters :935-959).
 * The position filters of :2102-2121 are cluster-independent and arrive folded into pos_use. */
int orc_exon_loglik(const hlala_exon_in* in, int long_read_mode, double* LL, int32_t* mism)
{
    double insertionP = long_read_mode ? 0.075 : 0.001, deletionP = insertionP;
    double log_likelihood_insertion = log(insertionP);
    double log_likelihood_insertion_actualAllele = log_likelihood_insertion + log(1.0 / 4.0);
    double log_likelihood_deletion = log(deletionP);
    double log_likelihood_match_mismatch = log(1 - insertionP - deletionP);
    const int C = in->n_clusters, P = in->exon_length, R = in->n_reads;
    /* (clusters are independent; every (cluster, read) sum below runs in the reference's order) */
#pragma omp parallel for schedule(dynamic, 8)
    for(int clusterI = 0; clusterI < C; clusterI++) {
        const unsigned char* clusterSequence = in->cluster_seq + (size_t)clusterI * P;
        for(int readI = 0; readI < R; readI++) {
            double log_likelihood_read = 0; int mismatches = 0;
            for(int i = in->pos_off[readI]; i < in->pos_off[readI + 1]; i++) {
                if(!in->pos_use[i]) continue;
                double log_likelihood_position = 0;
                char exonGenotype = (char)clusterSequence[in->pos_exon[i]];
                std::string readGenotype(1, (char)in->pos_g0[i]);
                readGenotype.append((size_t)(in->pos_glen[i] - 1), 'N');          /* inserted bases: only their count matters */
                unsigned int l_diff = (unsigned int)readGenotype.length() - 1;
                if(exonGenotype == '_') {
                    if(readGenotype == "_") { /* intrinsic graph gap, likelihood 1 */ }
                    else log_likelihood_position += (log_likelihood_insertion_actualAllele * (1 + l_diff));
                } else {
                    if(readGenotype[0] == '_') log_likelihood_position += log_likelihood_deletion;
                    else {
                        log_likelihood_position += log_likelihood_match_mismatch;
                        double pCorrect = PhredToPCorrect(in->pos_qual[i]);
                        if(pCorrect > 0.999) pCorrect = 0.999;
                        if(pCorrect == 0) pCorrect = 0.001;
                        if(exonGenotype == readGenotype[0]) log_likelihood_position += log(pCorrect);
                        else { double pIncorrect = (1 - pCorrect) * (1.0 / 3.0); log_likelihood_position += log(pIncorrect); }
                    }
                    log_likelihood_position += (log_likelihood_insertion_actualAllele * l_diff);
                }
                if(readGenotype != "_") if(readGenotype != std::string(1, exonGenotype)) mismatches++;
                log_likelihood_read += log_likelihood_position;
            }
            LL[(size_t)clusterI * R + readI] = log_likelihood_read;
            mism[(size_t)clusterI * R + readI] = mismatches;
        }
    }
    return 0;
}

/* Utilities::logAvg, Utilities.cpp:1368-1379 */
static double logAvg(double a, double b)
{
    if(a > b) return (log(0.5) + (log(1 + exp(b - a)) + a));
    return (log(0.5) + (log(1 + exp(a - b)) + b));
}

/* all cluster pairs in single-thread order, hla/HLATyper.cpp:2293-2364 */
int orc_pair_loglik(const double* LL, const int32_t* mism, int C, int R, double* pairLL, double* misAvg, double* misMin)
{
    /* (the reference walks c1, c2 on one thread; rows are independent and every sum below runs over the reads of ONE pair in the reference's order,
       so the rows may be computed side by side: pair (c1, c2) goes to the index the sequential walk gives it) */
#pragma omp parallel for schedule(dynamic, 4)
    for(int c1 = 0; c1 < C; c1++)
        for(int c2 = c1; c2 < C; c2++) {
            const size_t idx = (size_t)c1 * (size_t)C - (size_t)c1 * (size_t)(c1 - 1) / 2 + (size_t)(c2 - c1);
            double mismatches_sum_averages = 0, mismatches_sum_min = 0, pair_log_likelihood = 0;
            for(int readI = 0; readI < R; readI++) {
                double a = LL[(size_t)c1 * R + readI], b = LL[(size_t)c2 * R + readI];
                int m1 = mism[(size_t)c1 * R + readI], m2 = mism[(size_t)c2 * R + readI];
                mismatches_sum_averages += ((double)(m1 + m2) / 2.0);
                mismatches_sum_min += ((m1 < m2) ? m1 : m2);
                pair_log_likelihood += logAvg(a, b);
            }
            pairLL[idx] = pair_log_likelihood; misAvg[idx] = mismatches_sum_averages; misMin[idx] = mismatches_sum_min;
        }
    return 0;
}

// ---- exon positions of read pairs for one locus -------------------------------------------------------------------------
// One alignment as the typer sees it: columns of the selected chain + the read in alignment orientation.
struct TyperAln { int n; const int32_t* lv; const uint8_t* g; const uint8_t* s; const uint8_t* mq; const uint8_t* bases; const uint8_t* quals; int readLen; double mapQ; };
struct ExonPos { int positionInExon, graphLevel; std::string genotype, alignment_edgelabels, qualities; int mate; unsigned char mapqChar; int novelGap; };

static int aln_firstLevel(const TyperAln& a) { for(int i = 0; i < a.n; i++) if(a.lv[i] != -1) return a.lv[i]; return -1; }      // verboseSeedChain.h:122-135
static int aln_lastLevel(const TyperAln& a) { for(int i = a.n - 1; i >= 0; i--) if(a.lv[i] != -1) return a.lv[i]; return -1; }  // :157-183
// HLATyper::alignmentFractionOK, hla/HLATyper.cpp:3082-3101
static double alignmentFractionOK(const TyperAln& r)
{
    int positions_OK = 0, positions_checked = 0;
    for(int i = 0; i < r.n; i++) {
        if((r.g[i] == '_') && (r.s[i] == '_')) continue;
        positions_checked++;
        if(r.g[i] == r.s[i]) positions_OK++;
    }
    return double(positions_OK) / double(positions_checked);
}
// HLATyper::alignmentWeightedOKFraction, hla/HLATyper.cpp:3933-4018 (sequence_begin = 0 for extended chains; the reverse-index
// arithmetic of :3949-3959 lands on the alignment-orientation base, which is what `bases` / `quals` hold)
static int alignmentWeightedOKFraction(const TyperAln& a, double& out)
{
    int indexIntoOriginalReadData = 0 - 1;
    int totalMismatches = 0; double weightedMismatches = 0;
    for(int cI = 0; cI < a.n; cI++) {
        unsigned char sequenceCharacter = a.s[cI], graphCharacter = a.g[cI];
        if(sequenceCharacter != '_') {
            indexIntoOriginalReadData++;
            if(!(indexIntoOriginalReadData >= 0 && indexIntoOriginalReadData < a.readLen)) return -1;
            if(a.bases[indexIntoOriginalReadData] != sequenceCharacter) return -2;            // assert(underlyingReadCharacter == sequenceCharacter)
            if(graphCharacter == '_') { totalMismatches++; weightedMismatches++; }
            else {
                double pCorrect = PhredToPCorrect(a.quals[indexIntoOriginalReadData]);
                if(!((pCorrect >= 0) && (pCorrect <= 1))) return -3;
                if(sequenceCharacter != graphCharacter) { weightedMismatches += pCorrect; totalMismatches++; }
            }
        } else {
            if(graphCharacter == '_') { if(a.lv[cI] == -1) return -4; }
            else { totalMismatches++; weightedMismatches++; }
        }
    }
    double readLength = a.readLen;
    out = (1.0 - (weightedMismatches / readLength));
    return 0;
}
// alignerBase::alignedReadPair_pairsDistanceInGraphLevels, mapper/aligner/alignerBase.cpp:246-283
static int pairsDistanceInGraphLevels(const TyperAln& r1, const TyperAln& r2)
{
    if(aln_firstLevel(r1) < aln_firstLevel(r2)) return (aln_firstLevel(r2) - aln_lastLevel(r1) - 1);
    return (aln_firstLevel(r1) - aln_lastLevel(r2) - 1);
}
// HLATyper::oneReadAlignment_2_exonPositions_paired, hla/HLATyper.cpp:3192-3565 (only the fields that differ per position are kept)
static int oneReadAlignment_2_exonPositions(const TyperAln& alignment, int mate, std::vector<ExonPos>& ret, int levels_min, int levels_max, const int32_t* level_to_exon, int& colsNonGap)
{
    int alignment_firstLevel = aln_firstLevel(alignment), alignment_lastLevel = aln_lastLevel(alignment);
    colsNonGap = 0;
    if(!(alignment_firstLevel <= alignment_lastLevel)) return -10;
    if(!intervalsOverlap(alignment_firstLevel, alignment_lastLevel, levels_min, levels_max)) return 0;
    std::vector<ExonPos> readAlignment_exonPositions;
    int alignmentColumns_oneNonGap = 0;
    for(int cI = 0; cI < alignment.n; cI++) if((alignment.s[cI] != '_') || (alignment.s[cI] != '_')) alignmentColumns_oneNonGap++;      // sic, :3236
    colsNonGap = alignmentColumns_oneNonGap;
    std::vector<int> runningNovelGaps(alignment.n, 0);
    {
        int runningNovelGap = 0;
        for(int cI = 0; cI < alignment.n; cI++) {
            unsigned char sc = alignment.s[cI], gc = alignment.g[cI];
            if((gc != '_') && (sc != '_')) runningNovelGap = 0;
            else if(!((gc == '_') && (sc == '_'))) runningNovelGap++;
            if(runningNovelGap > runningNovelGaps.at(cI)) runningNovelGaps.at(cI) = runningNovelGap;
        }
        runningNovelGap = 0;
        for(int cI = alignment.n - 1; cI >= 0; cI--) {
            unsigned char sc = alignment.s[cI], gc = alignment.g[cI];
            if((gc != '_') && (sc != '_')) runningNovelGap = 0;
            else if(!((gc == '_') && (sc == '_'))) runningNovelGap++;
            if(runningNovelGap > runningNovelGaps.at(cI)) runningNovelGaps.at(cI) = runningNovelGap;
        }
    }
    int indexIntoOriginalReadData = -1;
    for(int cI = 0; cI < alignment.n; cI++) {
        unsigned char sequenceCharacter = alignment.s[cI], graphCharacter = alignment.g[cI];
        int graphLevel = alignment.lv[cI];
        if(graphLevel == -1) {
            // insertion relative to the graph - extend the last character, :3299-3340
            if(!(graphCharacter == '_') || !(sequenceCharacter != '_')) return -11;
            indexIntoOriginalReadData++;
            if(!(indexIntoOriginalReadData >= 0 && indexIntoOriginalReadData < alignment.readLen)) return -12;
            if(alignment.bases[indexIntoOriginalReadData] != sequenceCharacter) return -13;
            char qualityCharacter = (char)alignment.quals[indexIntoOriginalReadData];
            if(readAlignment_exonPositions.size() > 0) {
                ExonPos& bk = readAlignment_exonPositions.back();
                bk.genotype.push_back((char)sequenceCharacter); bk.alignment_edgelabels.push_back((char)graphCharacter); bk.qualities.push_back(qualityCharacter);
                if(!(bk.genotype.length() == bk.qualities.length())) {
                    if(!(bk.genotype.length() == (bk.qualities.length() + 1)) || !(bk.genotype.at(0) == '_')) return -14;
                    bk.genotype = bk.genotype.substr(1); bk.alignment_edgelabels = bk.alignment_edgelabels.substr(1);
                }
                if(bk.genotype.length() != bk.qualities.length()) return -15;
            }
        } else {
            ExonPos thisPosition;
            thisPosition.graphLevel = graphLevel; thisPosition.positionInExon = -1; thisPosition.mate = mate;
            thisPosition.mapqChar = alignment.mq[cI]; thisPosition.novelGap = runningNovelGaps.at(cI);
            thisPosition.alignment_edgelabels = std::string(1, (char)graphCharacter);
            if(sequenceCharacter != '_') {
                indexIntoOriginalReadData++;
                if(!(indexIntoOriginalReadData >= 0 && indexIntoOriginalReadData < alignment.readLen)) return -16;
                if(alignment.bases[indexIntoOriginalReadData] != sequenceCharacter) return -17;
                thisPosition.genotype = std::string(1, (char)sequenceCharacter);
                thisPosition.qualities = std::string(1, (char)alignment.quals[indexIntoOriginalReadData]);       // both branches :3360-3427
            } else {
                thisPosition.genotype = "_"; thisPosition.qualities = "";                                         // :3431-3490
            }
            readAlignment_exonPositions.push_back(thisPosition);
        }
    }
    int alongReadMode = 0, lastPositionInExon = -1;
    for(size_t posInAlignment = 0; posInAlignment < readAlignment_exonPositions.size(); posInAlignment++) {
        ExonPos& thisPosition = readAlignment_exonPositions.at(posInAlignment);
        bool inExon = thisPosition.graphLevel >= levels_min && thisPosition.graphLevel <= levels_max && level_to_exon[thisPosition.graphLevel - levels_min] >= 0;
        if(inExon) {
            if(alongReadMode == 2) lastPositionInExon = -1;
            thisPosition.positionInExon = level_to_exon[thisPosition.graphLevel - levels_min];
            if(!((lastPositionInExon == -1) || (thisPosition.positionInExon == (lastPositionInExon + 1)))) return -18;     // assert, :3528
            lastPositionInExon = thisPosition.positionInExon;
            alongReadMode = 1;
            ret.push_back(thisPosition);
        } else if(alongReadMode == 1) alongReadMode = 2;
    }
    return 0;
}
// HLATyper::removeDoublePositionsFromRead, hla/HLATyper.cpp:4020-4083
static int removeDoublePositionsFromRead(const std::vector<ExonPos>& positions, std::vector<ExonPos>& forReturn)
{
    std::map<int, std::vector<ExonPos>> positions_per_graphLevel;
    for(size_t i = 0; i < positions.size(); i++) positions_per_graphLevel[positions[i].graphLevel].push_back(positions[i]);
    for(auto it = positions_per_graphLevel.begin(); it != positions_per_graphLevel.end(); it++) {
        const std::vector<ExonPos>& alternatives = it->second;
        unsigned int bestI = 0; unsigned char bestI_quality = 0;
        for(unsigned int i = 0; i < alternatives.size(); i++) {
            const ExonPos& t = alternatives[i];
            if(!((t.genotype == "_") || (t.qualities.size() > 0))) return -20;
            unsigned char Q = 0;
            if(t.genotype != "_") { for(unsigned int k = 0; k < t.qualities.size(); k++) if((k == 0) || ((unsigned char)t.qualities[k] < Q)) Q = (unsigned char)t.qualities[k]; }
            if((i == 0) || (Q > bestI_quality)) { bestI = i; bestI_quality = Q; }
        }
        forReturn.push_back(alternatives.at(bestI));
    }
    return 0;
}

// The paired-read part of the per-locus loop of HLATyper::HLATypeInference, hla/HLATyper.cpp:1385-1428.
int orc_exon_positions(int n_pairs, int stride, const int32_t* pair_status, const int32_t* n_cols, const int32_t* col_level, const uint8_t* col_gchar,
                       const uint8_t* col_schar, const uint8_t* col_mapq, const double* mate_mapq, const uint8_t* strands_valid,
                       const int32_t* read_off, const uint8_t* read_bases, const uint8_t* read_quals,
                       const hlala_locus_desc* L, hlala_exon_positions_out* o)
{
    int nReads = 0, nPos = 0, nChars = 0; bool overflow = false;
    o->n_pairs_ok = 0; o->n_pairs_broken = 0;
    for(int p = 0; p < n_pairs; p++) {
        if(L->pair_mask && !L->pair_mask[p]) continue;
        if(pair_status[p] != 0) continue;
        TyperAln a[2];
        for(int m = 0; m < 2; m++) {
            size_t r = (size_t)2 * p + m;
            a[m].n = n_cols[r]; a[m].lv = col_level + r * stride; a[m].g = col_gchar + r * stride; a[m].s = col_schar + r * stride; a[m].mq = col_mapq + r * stride;
            a[m].bases = read_bases + read_off[r]; a[m].quals = read_quals + read_off[r]; a[m].readLen = read_off[r + 1] - read_off[r]; a[m].mapQ = mate_mapq[r];
        }
        std::vector<ExonPos> read1_exonPositions, read2_exonPositions; int cng[2] = {0, 0};
        int rc = oneReadAlignment_2_exonPositions(a[0], 1, read1_exonPositions, L->level_min, L->level_max, L->level_to_exon, cng[0]); if(rc) return rc;
        rc = oneReadAlignment_2_exonPositions(a[1], 2, read2_exonPositions, L->level_min, L->level_max, L->level_to_exon, cng[1]); if(rc) return rc;
        double w[2]; if((rc = alignmentWeightedOKFraction(a[0], w[0])) || (rc = alignmentWeightedOKFraction(a[1], w[1]))) return rc;
        int dist = pairsDistanceInGraphLevels(a[0], a[1]);
        double mapQ_thisAlignment = a[0].mapQ;
        if(!((mapQ_thisAlignment >= 0) && (mapQ_thisAlignment <= 1))) return -30;
        // `abs` on a double: Utilities.h:20 puts namespace std in scope, so the reference resolves to std::abs(double) (no truncation)
        if(strands_valid[p] && (std::abs(dist - L->insert_mean) <= (5 * L->insert_sd)) && (mapQ_thisAlignment >= L->min_mapq) && ((w[0] >= L->min_weighted_ok) && (w[1] >= L->min_weighted_ok))) {
            std::vector<ExonPos> thisRead_exonPositions = read1_exonPositions;
            thisRead_exonPositions.insert(thisRead_exonPositions.end(), read2_exonPositions.begin(), read2_exonPositions.end());
            if(thisRead_exonPositions.size() > 0) {
                std::vector<ExonPos> cleaned; if((rc = removeDoublePositionsFromRead(thisRead_exonPositions, cleaned))) return rc;
                size_t chars = 0; for(auto& e : cleaned) chars += e.genotype.size();
                if(nReads + 1 > o->cap_reads || nPos + (int)cleaned.size() > o->cap_pos || nChars + (int)chars > o->cap_chars) overflow = true;
                if(!overflow) {
                    o->read_pair[nReads] = p; o->read_weighted_ok[2 * nReads] = w[0]; o->read_weighted_ok[2 * nReads + 1] = w[1];
                    o->read_fraction_ok[2 * nReads] = alignmentFractionOK(a[0]); o->read_fraction_ok[2 * nReads + 1] = alignmentFractionOK(a[1]);
                    o->read_distance[nReads] = dist; o->read_cols_nongap[2 * nReads] = cng[0]; o->read_cols_nongap[2 * nReads + 1] = cng[1];
                    if(o->read_mapq) { o->read_mapq[2 * nReads] = a[0].mapQ; o->read_mapq[2 * nReads + 1] = a[1].mapQ; }      /* thisPosition.mapQ = alignment.mapQ, :3387 */
                    o->pos_off[nReads] = nPos;
                    int q = nPos, ch = nChars;
                    for(auto& e : cleaned) {
                        o->pos_exon[q] = e.positionInExon; o->pos_level[q] = e.graphLevel; o->pos_mate[q] = (uint8_t)e.mate; o->pos_mapq[q] = e.mapqChar; o->pos_novel_gap[q] = e.novelGap;
                        o->geno_off[q] = ch;
                        for(size_t k = 0; k < e.genotype.size(); k++) { o->geno_chars[ch] = (uint8_t)e.genotype[k]; o->qual_chars[ch] = k < e.qualities.size() ? (uint8_t)e.qualities[k] : 0; ch++; }
                        q++;
                    }
                }
                nReads++; nPos += (int)cleaned.size(); nChars += (int)chars;
            }
            o->n_pairs_ok++;
        } else o->n_pairs_broken++;
    }
    o->n_reads = nReads; o->n_pos = nPos; o->n_chars = nChars;
    if(overflow) return -100;
    o->pos_off[nReads] = nPos; o->geno_off[nPos] = nChars;
    return 0;
}

// The unpaired-read part of the per-locus loop, hla/HLATyper.cpp:1470-1493 with oneReadAlignment_2_exonPositions_unpaired (:3568-3930: the paired
// routine without the mate; pairedRead_* = -1, pairs_strands_distance = -1).  Arrays are per read ([n] where the paired layout has [2n]).
int orc_exon_positions_unpaired(int n_reads, int stride, const int32_t* pair_status, const int32_t* n_cols, const int32_t* col_level, const uint8_t* col_gchar,
                                const uint8_t* col_schar, const uint8_t* col_mapq, const double* mate_mapq,
                                const int32_t* read_off, const uint8_t* read_bases, const uint8_t* read_quals,
                                const hlala_locus_desc* L, hlala_exon_positions_out* o)
{
    int nReads = 0, nPos = 0, nChars = 0; bool overflow = false;
    o->n_pairs_ok = 0; o->n_pairs_broken = 0;
    for(int r = 0; r < n_reads; r++) {
        if(L->pair_mask && !L->pair_mask[r]) continue;
        if(pair_status[r] != 0) continue;
        TyperAln a;
        a.n = n_cols[r]; a.lv = col_level + (size_t)r * stride; a.g = col_gchar + (size_t)r * stride; a.s = col_schar + (size_t)r * stride; a.mq = col_mapq + (size_t)r * stride;
        a.bases = read_bases + read_off[r]; a.quals = read_quals + read_off[r]; a.readLen = read_off[r + 1] - read_off[r]; a.mapQ = mate_mapq[r];
        std::vector<ExonPos> read_exonPositions; int cng = 0;
        int rc = oneReadAlignment_2_exonPositions(a, 1, read_exonPositions, L->level_min, L->level_max, L->level_to_exon, cng); if(rc) return rc;
        double w; if((rc = alignmentWeightedOKFraction(a, w))) return rc;
        double mapQ_thisAlignment = a.mapQ;
        if((mapQ_thisAlignment >= L->min_mapq) && (a.n >= L->min_alignment_columns)) {               // :1476
            if(read_exonPositions.size() > 0) {                                                       // pushed as they are: no removeDoublePositionsFromRead, :1481-1485
                size_t chars = 0; for(auto& e : read_exonPositions) chars += e.genotype.size();
                if(nReads + 1 > o->cap_reads || nPos + (int)read_exonPositions.size() > o->cap_pos || nChars + (int)chars > o->cap_chars) overflow = true;
                if(!overflow) {
                    o->read_pair[nReads] = r; o->read_weighted_ok[2 * nReads] = w; o->read_weighted_ok[2 * nReads + 1] = -1;
                    o->read_fraction_ok[2 * nReads] = alignmentFractionOK(a); o->read_fraction_ok[2 * nReads + 1] = -1;
                    o->read_distance[nReads] = -1; o->read_cols_nongap[2 * nReads] = cng; o->read_cols_nongap[2 * nReads + 1] = 0;
                    if(o->read_mapq) { o->read_mapq[2 * nReads] = a.mapQ; o->read_mapq[2 * nReads + 1] = -1; }
                    o->pos_off[nReads] = nPos;
                    int q = nPos, ch = nChars;
                    for(auto& e : read_exonPositions) {
                        o->pos_exon[q] = e.positionInExon; o->pos_level[q] = e.graphLevel; o->pos_mate[q] = 1; o->pos_mapq[q] = e.mapqChar; o->pos_novel_gap[q] = e.novelGap;
                        o->geno_off[q] = ch;
                        for(size_t k = 0; k < e.genotype.size(); k++) { o->geno_chars[ch] = (uint8_t)e.genotype[k]; o->qual_chars[ch] = k < e.qualities.size() ? (uint8_t)e.qualities[k] : 0; ch++; }
                        q++;
                    }
                }
                nReads++; nPos += (int)read_exonPositions.size(); nChars += (int)chars;
            }
            o->n_pairs_ok++;
        } else o->n_pairs_broken++;
    }
    o->n_reads = nReads; o->n_pos = nPos; o->n_chars = nChars;
    if(overflow) return -100;
    o->pos_off[nReads] = nPos; o->geno_off[nPos] = nChars;
    return 0;
}

// Read / allele filters of HLATyper::HLATypeInference: filterFirst20 (hla/HLATyper.cpp:1496-1720), the high-coverage allele filter
// (:1722-1862, short reads: the strand filter of :1847-1858 needs longReadsMode) and the use test of the likelihood loop (:2102-2120).
// Reads are the entries of exonPositions_fromReads; a read's ID pair is unique to the entry, so ignore_readIDs is a set of entries.
int orc_filter_positions(const hlala_exon_positions_out* pos, const hlala_filter_params* prm, uint8_t* pos_use, uint8_t* read_ignored, hlala_filter_stats* st)
{
    const int nReads = pos->n_reads;
    auto genotype = [&](int j) { return std::string((const char*)pos->geno_chars + pos->geno_off[j], (size_t)(pos->geno_off[j + 1] - pos->geno_off[j])); };
    auto mapQ_position = [&](int j) { return PhredToPCorrect(pos->pos_mapq[j]); };
    auto weightedOK = [&](int readI) { return (pos->read_weighted_ok[2 * readI] + pos->read_weighted_ok[2 * readI + 1]) / 2.0; };   // commutative: same for both mates
    std::set<int> ignore_readIDs;
    std::map<unsigned int, std::set<std::string>> perPosition_ignore_alleles;
    hlala_filter_stats S; memset(&S, 0, sizeof(S));
    const bool filterFirst20 = prm->filter_first20 != 0;
    if(filterFirst20) {
        std::map<unsigned int, int> perRead_kickedOut, perRead_kickedOut_robust;
        std::map<unsigned int, std::vector<std::string>> perPosition_alleles;
        std::map<unsigned int, std::vector<double>> perPosition_weightedOK;
        std::map<unsigned int, std::vector<unsigned int>> perPosition_readI;
        for(int readI = 0; readI < nReads; readI++)
            for(int j = pos->pos_off[readI]; j < pos->pos_off[readI + 1]; j++) {
                if(!((mapQ_position(j) >= 0) && (mapQ_position(j) <= 1))) return -1;
                if(mapQ_position(j) < prm->min_per_position_mapq) continue;
                int position = pos->pos_exon[j];
                perPosition_alleles[position].push_back(genotype(j));
                perPosition_weightedOK[position].push_back(weightedOK(readI));
                perPosition_readI[position].push_back((unsigned)readI);
            }
        for(auto position : perPosition_alleles) {
            int n_alleles = (int)perPosition_alleles.at(position.first).size();
            if(n_alleles < prm->first20_n) continue;
            std::vector<unsigned int> allele_indices; allele_indices.reserve(n_alleles);
            for(unsigned int i = 0; i < (unsigned)n_alleles; i++) allele_indices.push_back(i);
            std::sort(allele_indices.begin(), allele_indices.end(), [&](unsigned int a, unsigned int b) {
                return (perPosition_weightedOK.at(position.first).at(a) < perPosition_weightedOK.at(position.first).at(b)); });
            std::reverse(allele_indices.begin(), allele_indices.end());
            std::map<std::string, int> alleles_first20;
            std::set<std::string> kickedOutAlleles;
            for(unsigned int i = 0; i < (unsigned)prm->first20_n; i++) {
                std::string allele = perPosition_alleles.at(position.first).at(allele_indices.at(i));
                if(alleles_first20.count(allele) == 0) alleles_first20[allele] = 0;
                alleles_first20.at(allele)++;
            }
            bool kickedOneOut = false;
            for(unsigned int i = 0; i < (unsigned)n_alleles; i++) {
                std::string allele = perPosition_alleles.at(position.first).at(i);
                unsigned int readI = perPosition_readI.at(position.first).at(i);
                int first20_alleleCount = alleles_first20.count(allele) ? alleles_first20.at(allele) : 0;
                double first20_prop = (double)first20_alleleCount / (double)filterFirst20;        // sic: the bool, not filterFirst20N (:1593)
                S.considered_alleles++;
                if(first20_prop < prm->first20_min_prop) {
                    kickedOutAlleles.insert(allele); perPosition_ignore_alleles[position.first].insert(allele);
                    if(perRead_kickedOut.count(readI) == 0) perRead_kickedOut[readI] = 0;
                    perRead_kickedOut.at(readI)++; kickedOneOut = true; S.removed_alleles++;
                }
            }
            std::map<std::string, int> allele_kickedOut_howMany;
            for(unsigned int i = 0; i < (unsigned)n_alleles; i++) {
                std::string allele = perPosition_alleles.at(position.first).at(i);
                if(kickedOutAlleles.count(allele)) { if(allele_kickedOut_howMany.count(allele) == 0) allele_kickedOut_howMany[allele] = 0; allele_kickedOut_howMany.at(allele)++; }
            }
            for(unsigned int i = 0; i < (unsigned)n_alleles; i++) {
                std::string allele = perPosition_alleles.at(position.first).at(i);
                unsigned int readI = perPosition_readI.at(position.first).at(i);
                if(allele_kickedOut_howMany.count(allele) && (allele_kickedOut_howMany.at(allele) >= 2)) {
                    if(perRead_kickedOut_robust.count(readI) == 0) perRead_kickedOut_robust[readI] = 0;
                    perRead_kickedOut_robust.at(readI)++;
                }
            }
            S.considered_positions++;
            if(kickedOneOut) S.positions_with_removed_alleles++;
        }
        for(auto readKickedOut : perRead_kickedOut) if(readKickedOut.second > prm->first20_limit_per_read) S.reads_kicked_out++;     // counted only, :1661-1668
        for(auto readKickedOut : perRead_kickedOut_robust)
            if(readKickedOut.second > prm->first20_limit_per_read) { S.reads_kicked_out_robust++; ignore_readIDs.insert((int)readKickedOut.first); }   // :1682-1690
    }
    {   // high-coverage allele filter, :1722-1862
        std::map<unsigned int, std::map<std::string, int>> perPosition_allele_counts;
        std::map<unsigned int, std::map<std::string, std::pair<int, int>>> perPosition_allele_counts_byStrand;
        for(int readI = 0; readI < nReads; readI++)
            for(int j = pos->pos_off[readI]; j < pos->pos_off[readI + 1]; j++) {
                if(ignore_readIDs.count(readI)) continue;
                if(mapQ_position(j) < prm->min_per_position_mapq) continue;
                int position = pos->pos_exon[j]; std::string allele = genotype(j);
                if(perPosition_ignore_alleles.count(position) && (perPosition_ignore_alleles.at(position).count(allele))) continue;
                if(perPosition_allele_counts[position].count(allele) == 0) { perPosition_allele_counts[position][allele] = 0; perPosition_allele_counts_byStrand[position][allele] = std::make_pair(0, 0); }
                perPosition_allele_counts.at(position).at(allele)++;
                if(pos->read_reverse) {                                                           /* onePositionSpecifier.reverse, :1768-1775 */
                    bool reverse = pos->read_reverse[2 * readI + (pos->pos_mate[j] == 2 ? 1 : 0)] != 0;
                    if(reverse) perPosition_allele_counts_byStrand.at(position).at(allele).second++; else perPosition_allele_counts_byStrand.at(position).at(allele).first++;
                }
            }
        for(auto position : perPosition_allele_counts) {
            int count_position = 0;
            for(auto allele : perPosition_allele_counts.at(position.first)) count_position += allele.second;
            if(count_position >= prm->high_coverage_min_coverage) {
                S.high_coverage_positions++;
                for(auto allele : perPosition_allele_counts.at(position.first)) {
                    double aF = (double)allele.second / (double)count_position;
                    if((aF < prm->high_coverage_min_freq) && prm->high_coverage_filter) { perPosition_ignore_alleles[position.first].insert(allele.first); S.high_coverage_removed_alleles += allele.second; }
                }
            }
            bool kickedOutAtLeastOneAllele_sF = false;                                            /* :1826-1861 */
            if(pos->read_reverse) for(auto allele : perPosition_allele_counts_byStrand.at(position.first)) {
                int totalCount = allele.second.first + allele.second.second;
                int minStrandCount = (allele.second.first < allele.second.second) ? allele.second.first : allele.second.second;
                double minStrandFreq = (double)minStrandCount / (double)totalCount;
                if(prm->long_read_strand_filter && (totalCount >= prm->strand_min_allele_coverage)) {
                    S.strand_alleles_enough_coverage++;
                    if(minStrandFreq < prm->strand_min_freq) { perPosition_ignore_alleles[position.first].insert(allele.first); S.strand_removed_alleles++; kickedOutAtLeastOneAllele_sF = true; }
                }
                if(kickedOutAtLeastOneAllele_sF) S.strand_positions_with_removed++;
            }
        }
    }
    for(int readI = 0; readI < nReads; readI++) {
        if(read_ignored) read_ignored[readI] = ignore_readIDs.count(readI) ? 1 : 0;
        for(int j = pos->pos_off[readI]; j < pos->pos_off[readI + 1]; j++) {
            bool use = true;                                                                      // :2102-2120
            if(mapQ_position(j) < prm->min_per_position_mapq) use = false;
            else if(perPosition_ignore_alleles.count(pos->pos_exon[j]) && (perPosition_ignore_alleles.at(pos->pos_exon[j]).count(genotype(j)))) use = false;
            else if(ignore_readIDs.count(readI)) use = false;
            pos_use[j] = use ? 1 : 0;
            if(use) S.bases_used++;
        }
    }
    if(st) *st = S;
    return 0;
}

// The call of one locus (hla/HLATyper.cpp:2366-2541) from the all-pairs table; same index convention as orc_pair_loglik
// (LLs_clusterIs is filled c1-major, c2 >= c1, :2293-2364).  std::sort + std::reverse are the reference's own calls (:2381-2403).
int orc_call_locus(int C, const double* pairLL, const double* misAvg, const double* misMin, int32_t* order, double* p_normalized,
                   double* cluster_marginal, hlala_call_out* out)
{
    const size_t nP = (size_t)C * (C + 1) / 2;
    if(C < 1) return -1;
    std::vector<std::pair<unsigned int, unsigned int>> LLs_clusterIs; LLs_clusterIs.reserve(nP);
    for(int c1 = 0; c1 < C; c1++) for(int c2 = c1; c2 < C; c2++) LLs_clusterIs.push_back(std::make_pair((unsigned)c1, (unsigned)c2));
    std::vector<double> LLs_completeReads(pairLL, pairLL + nP), Mismatches_avg(misAvg, misAvg + nP), Mismatches_min(misMin, misMin + nP);
    std::vector<size_t> LLs_completeReads_indices;
    for(size_t i = 0; i < LLs_completeReads.size(); i++) LLs_completeReads_indices.push_back(i);
    std::sort(LLs_completeReads_indices.begin(), LLs_completeReads_indices.end(), [&](unsigned int a, unsigned int b) {
        if(LLs_completeReads.at(a) == LLs_completeReads.at(b)) return (Mismatches_avg.at(b) < Mismatches_avg.at(a));
        else return (LLs_completeReads.at(a) < LLs_completeReads.at(b));
    });
    std::reverse(LLs_completeReads_indices.begin(), LLs_completeReads_indices.end());
    // findVectorMax, Utilities.cpp:309-324: first maximum
    double LL_max = 0; unsigned iMax = 0;
    for(unsigned int i = 0; i < LLs_completeReads.size(); i++) if((i == 0) || (LLs_completeReads.at(i) > LL_max)) { iMax = i; LL_max = LLs_completeReads.at(i); }
    std::vector<double> LLs_normalized;
    double P_sum = 0;
    for(unsigned int cI = 0; cI < LLs_clusterIs.size(); cI++) P_sum += exp(LLs_completeReads.at(cI) - LL_max);
    if(P_sum > 0) {
        for(unsigned int cI = 0; cI < LLs_clusterIs.size(); cI++) {
            double P_normalized = exp(LLs_completeReads.at(cI) - LL_max) / P_sum;
            if(!((P_normalized >= 0) && (P_normalized <= 1))) return -2;      // assert, :2438-2439
            LLs_normalized.push_back(P_normalized);
        }
    } else for(unsigned int cI = 0; cI < LLs_clusterIs.size(); cI++) LLs_normalized.push_back(1.0 / (double)LLs_clusterIs.size());
    std::map<int, double> clusterI_overAllPairs;
    for(unsigned int cII = 0; cII < LLs_completeReads_indices.size(); cII++) {
        unsigned int cI = LLs_completeReads_indices.at(cII);
        std::pair<unsigned int, unsigned int>& clusters = LLs_clusterIs.at(cI);
        if(clusterI_overAllPairs.count(clusters.first) == 0) clusterI_overAllPairs[clusters.first] = 0;
        clusterI_overAllPairs[clusters.first] += LLs_normalized.at(cI);
        if(clusters.second != clusters.first) {
            if(clusterI_overAllPairs.count(clusters.second) == 0) clusterI_overAllPairs[clusters.second] = 0;
            clusterI_overAllPairs[clusters.second] += LLs_normalized.at(cI);
        }
    }
    auto findIntMapMax = [](std::map<int, double>& m) {                         // Utilities.cpp:257-272: first maximum in key order
        double max = 0; int iMaxK = 0;
        for(std::map<int, double>::iterator mIt = m.begin(); mIt != m.end(); mIt++) if((mIt == m.begin()) || (mIt->second > max)) { max = mIt->second; iMaxK = mIt->first; }
        return std::pair<double, int>(max, iMaxK);
    };
    std::pair<double, int> bestGuess_firstAllele = findIntMapMax(clusterI_overAllPairs);
    std::map<int, double> bestGuess_secondAllele_alternatives, bestGuess_secondAllele_alternatives_mismatches;
    for(unsigned int cI = 0; cI < LLs_clusterIs.size(); cI++) {
        std::pair<unsigned int, unsigned int>& clusters = LLs_clusterIs.at(cI);
        if((int)clusters.first == bestGuess_firstAllele.second) {
            bestGuess_secondAllele_alternatives[clusters.second] = LLs_normalized.at(cI);
            bestGuess_secondAllele_alternatives_mismatches[clusters.second] = Mismatches_min.at(cI);
        } else if((int)clusters.second == bestGuess_firstAllele.second) {
            bestGuess_secondAllele_alternatives[clusters.first] = LLs_normalized.at(cI);
            bestGuess_secondAllele_alternatives_mismatches[clusters.first] = Mismatches_min.at(cI);
        }
    }
    std::pair<double, int> oneBestGuess_secondAllele = findIntMapMax(bestGuess_secondAllele_alternatives);
    std::map<int, double> mismatches_allBestGuessPairs;
    for(std::map<int, double>::iterator it = bestGuess_secondAllele_alternatives.begin(); it != bestGuess_secondAllele_alternatives.end(); it++)
        if(it->second == oneBestGuess_secondAllele.first) mismatches_allBestGuessPairs[it->first] = -1 * bestGuess_secondAllele_alternatives_mismatches.at(it->first);
    std::pair<double, int> bestGuess_secondAllele = findIntMapMax(mismatches_allBestGuessPairs);
    if(order) for(size_t i = 0; i < nP; i++) order[i] = (int32_t)LLs_completeReads_indices[i];
    if(p_normalized) for(size_t i = 0; i < nP; i++) p_normalized[i] = LLs_normalized[i];
    if(cluster_marginal) for(int c = 0; c < C; c++) cluster_marginal[c] = clusterI_overAllPairs.count(c) ? clusterI_overAllPairs.at(c) : 0.0;
    if(out) {
        out->first_cluster = bestGuess_firstAllele.second; out->second_cluster = bestGuess_secondAllele.second;
        out->first_marginal = bestGuess_firstAllele.first; out->second_p = oneBestGuess_secondAllele.first;
        out->ll_max = LL_max; out->max_pair = (int32_t)iMax;
        int ties = 0;
        for(size_t i = 1; i < nP; i++) { size_t a = LLs_completeReads_indices[i - 1], b = LLs_completeReads_indices[i]; if(pairLL[a] == pairLL[b] && misAvg[a] == misAvg[b]) ties++; }
        out->n_sort_ties = ties;
    }
    return 0;
}

/* PRGContigAlignment2Seed over column alignments handed in as seeds_in columns WITHOUT edges
 * (the simulateBAMAlignments route of --action testAlignments2Chains / testChainExtension,
 * HLA-LA.cpp:1622-1861): seq_begin/seq_end play sequence_aligned_{start,stop}InRaw. */
// Per-pair post-processing of alignReads_postSeedExtraction_andStoreInto (mapper/processBAM.cpp:2411-2446) on the selected
// chains of a batch: bases_per_level counters (:2411-2428) and includeInHLA (:2430-2446) through
// HLATyper::intervalOverlapsWithGenes (hla/HLATyper.cpp:259-268) = "some gene interval with stop >= first && start <= last"
// (IntervalTree::findOverlapping, intervalTree/IntervalTree.h:162-180: the tree only prunes, the test is the closed one at :166).
int orc_postprocess_pairs(int n_pairs, int stride, const int32_t* pair_status, const int32_t* n_cols /* [2n] */, const int32_t* col_level,
                          const uint8_t* col_gchar, int n_genes, const int32_t* gene_first, const int32_t* gene_last,
                          int n_cov, int32_t* bases_per_level /* in/out */, uint8_t* include_in_hla)
{
    for(int p = 0; p < n_pairs; p++) {
        bool includeInHLA = false;
        if(pair_status[p] == 0) {
            for(int m = 0; m < 2; m++) {
                const size_t r = (size_t)2 * p + m;
                const int32_t* lv = col_level + r * stride; const uint8_t* g = col_gchar + r * stride;
                const int n = n_cols[r];
                for(int aI = 0; aI < n; aI++) {
                    int level = lv[aI];
                    if((level != -1) && (g[aI] != '_')) { if(level < 0 || level >= n_cov) return -1; bases_per_level[level]++; }
                }
                int firstLevel = -1, lastLevel = -1;                       // verboseSeedChain::alignment_firstLevel / alignment_lastLevel, verboseSeedChain.h:122-183
                for(int i = 0; i < n; i++) if(lv[i] != -1) { firstLevel = lv[i]; break; }
                for(int i = n - 1; i >= 0; i--) if(lv[i] != -1) { lastLevel = lv[i]; break; }
                if(firstLevel != -1) {
                    if(!(firstLevel <= lastLevel)) return -2;              // assert, :2438
                    bool found = false;
                    for(int gI = 0; gI < n_genes; gI++) if(gene_last[gI] >= firstLevel && gene_first[gI] <= lastLevel) found = true;
                    includeInHLA = includeInHLA || found;
                }
            }
        }
        if(include_in_hla) include_in_hla[p] = includeInHLA ? 1 : 0;
    }
    return 0;
}

int orc_rethread_columns(orc_handle* h, const hlala_seeds_in* in, int restrict_gaps, hlala_chains_out* out)
{
    try {
        Processor& P = h->P;
        int stride = P.params.max_columns;
        for(int c = 0; c < in->n_chains; c++) {
            ContigAlignment ca; ca.startInRaw = in->chain_seq_begin[c]; ca.stopInRaw = in->chain_seq_end[c]; ca.reverse = in->chain_reverse[c] != 0;
            for(int j = in->col_off[c]; j < in->col_off[c + 1]; j++) {
                ca.levels.push_back(in->col_level[j]); ca.graph_aligned.push_back((char)in->col_gchar[j]); ca.sequence_aligned.push_back((char)in->col_schar[j]);
            }
            Chain s = P.PRGContigAlignment2Seed(ca, restrict_gaps != 0);
            storeChain(s, c, stride, out, HLALA_CHAIN_OK);
        }
        return 0;
    } catch(std::exception& e) { g_err = e.what(); return -1; }
}

} /* extern "C" */
