// C++ caller of the host mirror (hla-la_amd/host/hlala_host.hpp): the hand-derived exact-match extension case of
// tests/test_oracle_kat.py, written the way the reference's --action testChainExtension drives extendSeedChain
// (HLA-LA.cpp:1815-1835).  Needs a GPU to run; compiles anywhere.
#include <cassert>
#include <cstdio>
#include <string>

#include "../../hla-la_amd/host/hlala_host.hpp"

using namespace hlala::host;

int main()
{
    const std::string g = "ACGTACGTACGTACGTACGT";
    Graph graph; graph.n_levels = (int)g.size() + 1;
    for(int l = 0; l <= (int)g.size(); l++) graph.node_level.push_back(l);
    for(int l = 0; l < (int)g.size(); l++) { graph.edge_from.push_back(l); graph.edge_to.push_back(l + 1); graph.edge_label.push_back((uint8_t)g[l]); }
    try {
        mapper::aligner::extensionAligner g_extensionAligner(graph, nullptr, 200.0, 35.0);
        const std::string originalSequence = g.substr(4, 12);
        mapper::reads::verboseSeedChain sequenceSeed;                 // read[3..8] on levels 7..12
        sequenceSeed.sequence_begin = 3; sequenceSeed.sequence_end = 8;
        for(int i = 0; i < 6; i++) { sequenceSeed.graph_aligned_levels.push_back(7 + i); sequenceSeed.graph_aligned_edges.push_back(7 + i); }
        sequenceSeed.graph_aligned = originalSequence.substr(3, 6); sequenceSeed.sequence_aligned = originalSequence.substr(3, 6);
        mapper::reads::verboseSeedChain sequenceSeed_extended = g_extensionAligner.extendSeedChain(originalSequence, sequenceSeed);
        std::string noGaps;
        for(char c : sequenceSeed_extended.sequence_aligned) if(c != '_') noGaps.push_back(c);
        if(noGaps != originalSequence) { std::printf("FAIL: extended chain does not re-spell the read\n"); return 1; }    // HLA-LA.cpp:1824-1835
        if(sequenceSeed_extended.graph_aligned_levels.front() != 4 || sequenceSeed_extended.graph_aligned_levels.back() != 15) { std::printf("FAIL: levels\n"); return 1; }
        if(!(g_extensionAligner.scoreOneAlignment(sequenceSeed_extended, mapper::reads::oneRead()) < 0)) { std::printf("FAIL: LL\n"); return 1; }
        std::printf("HOST MIRROR OK %s %d..%d\n", sequenceSeed_extended.graph_aligned.c_str(), sequenceSeed_extended.graph_aligned_levels.front(), sequenceSeed_extended.graph_aligned_levels.back());
    } catch(std::exception& e) {
        std::printf("EXCEPTION %s\n", e.what());
        return 2;
    }
    return 0;
}
