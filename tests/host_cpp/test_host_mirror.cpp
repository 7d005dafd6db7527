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
        // processBAM::alignOneReadPair through alignReadPairs: a 120-level linear graph with its sequence as the one contig, a pair of
        // 30-base reads (mate 1 forward at 10, mate 2 reverse at 70; bases in alignment orientation), full-length matches
        {
            std::string ref; unsigned x = 12345u;
            for(int i = 0; i < 120; i++) { x = x * 1103515245u + 12345u; ref.push_back("ACGT"[(x >> 16) & 3]); }
            Graph g2; g2.n_levels = 121;
            for(int l = 0; l <= 120; l++) g2.node_level.push_back(l);
            for(int l = 0; l < 120; l++) { g2.edge_from.push_back(l); g2.edge_to.push_back(l + 1); g2.edge_label.push_back((uint8_t)ref[l]); }
            Contigs contigs; std::vector<int32_t> lv; for(int l = 0; l < 120; l++) lv.push_back(l);
            contigs.add(1, ref, lv);
            mapper::aligner::extensionAligner eA2(g2, &contigs, 30.0, 10.0);
            mapper::reads::protoSeeds ps; ps.readID = "pair1";
            mapper::reads::BamRecord r1; r1.contig = 0; r1.Position = 10; r1.AS = 30; r1.IsPrimaryAlignment = true; r1.CigarData.push_back((30u << 4) | 0u);
            mapper::reads::BamRecord r2 = r1; r2.Position = 70; r2.IsReverseStrand = true;
            ps.read1_alignments.push_back(r1); ps.read2_alignments.push_back(r2);
            ps.read1_QueryBases = ref.substr(10, 30); ps.read2_QueryBases = ref.substr(70, 30); ps.read1_Qualities = std::string(30, 'I'); ps.read2_Qualities = std::string(30, 'I');
            std::vector<mapper::reads::verboseSeedChainPair> al = eA2.alignReadPairs(std::vector<mapper::reads::protoSeeds>(1, ps));
            if(al.size() != 1 || al[0].chains.first.graph_aligned != ref.substr(10, 30) || al[0].chains.second.sequence_aligned != ref.substr(70, 30) ||
               al[0].chains.first.alignment_firstLevel() != 10 || al[0].chains.second.alignment_lastLevel() != 99 || !al[0].chains.second.reverse || al[0].chains.first.reverse ||
               al[0].chains.first.mapQ_perPosition.size() != 30 || !(al[0].mapQ > 0.99)) { std::printf("FAIL: alignReadPairs\n"); return 1; }
        }
        std::printf("HOST MIRROR OK %s %d..%d\n", sequenceSeed_extended.graph_aligned.c_str(), sequenceSeed_extended.graph_aligned_levels.front(), sequenceSeed_extended.graph_aligned_levels.back());
    } catch(std::exception& e) {
        std::printf("EXCEPTION %s\n", e.what());
        return 2;
    }
    return 0;
}
