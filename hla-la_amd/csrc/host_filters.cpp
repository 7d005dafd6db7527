// host_filters.cpp -- read / allele filters between the exon positions and the likelihoods (hla/HLATyper.cpp:1496-1862, :2102-2120).
//
// Host code by necessity: the "first 20" of an exon position are the first filterFirst20N entries after std::sort (+ std::reverse)
// of the reads' weighted-OK fractions (:1557-1565), and those fractions tie all the time (1.0 for every clean read), so the outcome
// is whatever order the C++ library's sort leaves tied keys in.  Running the same std::sort on the same index sequence reproduces the
// reference; no reformulation for the GPU can.  Everything else is flat arrays: positions are bucketed by exon position with a
// counting pass (read order preserved inside a bucket), alleles are numbered per bucket.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/hlala_gpu.h"
#include "host_internal.h"

namespace {

// Utilities::PhredToPCorrect, Utilities.cpp:357-377
double phred_to_pcorrect(unsigned char q)
{
    if(q == 0) return -1;
    double illuminaPhred = (double)q - 33;
    return 1 - exp(log(10.0) * (illuminaPhred / -10.0));
}

struct Entry { int pos_index; int read; int allele; double w; };     // one position of one read inside its exon-position bucket

}  // namespace

extern "C" int hlala_filter_positions(const hlala_exon_positions_out* pos, const hlala_filter_params* prm, uint8_t* pos_use, uint8_t* read_ignored, hlala_filter_stats* stats)
{
    try { return hlala_host::filter_positions_impl(pos, prm, pos_use, read_ignored, stats, nullptr); }
    catch(const std::exception&) { return HLALA_E_ARG; }      // (bad_alloc on absurd sizes: nothing may cross the C boundary)
}

int hlala_host::filter_positions_impl(const hlala_exon_positions_out* pos, const hlala_filter_params* prm, uint8_t* pos_use, uint8_t* read_ignored, hlala_filter_stats* stats,
                                      std::vector<std::vector<AlleleTally>>* tallies)
{
    if(!pos || !prm || !pos_use) return HLALA_E_ARG;
    const int nReads = pos->n_reads, nPos = pos->n_pos;
    hlala_filter_stats S; memset(&S, 0, sizeof(S));
    if(nReads > 0 && (!pos->pos_off || !pos->pos_exon || !pos->pos_mapq || !pos->geno_off || !pos->geno_chars || !pos->read_weighted_ok)) return HLALA_E_ARG;
    double pc[256]; for(int q = 0; q < 256; q++) pc[q] = phred_to_pcorrect((unsigned char)q);
    std::vector<uint8_t> mapqOK((size_t)nPos, 0);
    int maxExon = -1;
    for(int j = 0; j < nPos; j++) {
        const double m = pc[pos->pos_mapq[j]];
        if(!((m >= 0) && (m <= 1))) return HLALA_E_ARG;                                            // assert, :1526
        mapqOK[j] = m >= prm->min_per_position_mapq;
        if(pos->pos_exon[j] < 0) return HLALA_E_ARG;
        if(pos->pos_exon[j] > maxExon) maxExon = pos->pos_exon[j];
    }
    // ---- buckets by exon position, entries in (read, position) order; alleles numbered per bucket in order of first appearance
    std::vector<int> bOff((size_t)maxExon + 2, 0);
    for(int j = 0; j < nPos; j++) if(mapqOK[j]) bOff[(size_t)pos->pos_exon[j] + 1]++;
    for(int e = 0; e <= maxExon; e++) bOff[(size_t)e + 1] += bOff[e];
    std::vector<Entry> ent((size_t)bOff[(size_t)maxExon + 1]);
    std::vector<int> alleleOf((size_t)nPos, -1);
    {
        std::vector<int> fill(bOff.begin(), bOff.end() - 1);
        for(int r = 0; r < nReads; r++) {
            const double w = (pos->read_weighted_ok[2 * r] + pos->read_weighted_ok[2 * r + 1]) / 2.0;      // completeRead_weightedCharactersOK, :1535
            for(int j = pos->pos_off[r]; j < pos->pos_off[r + 1]; j++) if(mapqOK[j]) { Entry& E = ent[(size_t)fill[pos->pos_exon[j]]++]; E.pos_index = j; E.read = r; E.allele = -1; E.w = w; }
        }
    }
    std::vector<int> nAllelesOf((size_t)maxExon + 1, 0);
    {
        std::unordered_map<std::string, int> ids;
        for(int e = 0; e <= maxExon; e++) {
            ids.clear();
            for(int k = bOff[e]; k < bOff[(size_t)e + 1]; k++) {
                const int j = ent[k].pos_index;
                std::string a((const char*)pos->geno_chars + pos->geno_off[j], (size_t)(pos->geno_off[j + 1] - pos->geno_off[j]));
                auto it = ids.find(a);
                if(it == ids.end()) it = ids.emplace(a, (int)ids.size()).first;
                ent[k].allele = it->second; alleleOf[j] = it->second;
            }
            nAllelesOf[e] = (int)ids.size();
        }
    }
    // ignored alleles: flag per (bucket, allele id)
    std::vector<int> aOff((size_t)maxExon + 2, 0);
    for(int e = 0; e <= maxExon; e++) aOff[(size_t)e + 1] = aOff[e] + nAllelesOf[e];
    std::vector<uint8_t> ignoreAllele((size_t)aOff[(size_t)maxExon + 1], 0);
    std::vector<uint8_t> ignoreRead((size_t)nReads, 0);

    // ---- filterFirst20, :1509-1720
    if(prm->filter_first20) {
        std::vector<int> kicked((size_t)nReads, 0), kickedRobust((size_t)nReads, 0);
        std::vector<unsigned int> idx; std::vector<int> first20, kickedCount;
        for(int e = 0; e <= maxExon; e++) {
            const int b0 = bOff[e], n = bOff[(size_t)e + 1] - b0;
            if(n == 0 || n < prm->first20_n) continue;
            idx.resize((size_t)n); for(int i = 0; i < n; i++) idx[i] = (unsigned)i;
            const Entry* E = &ent[(size_t)b0];
            std::sort(idx.begin(), idx.end(), [&](unsigned int a, unsigned int b) { return E[a].w < E[b].w; });            // the reference's call, :1557-1563
            std::reverse(idx.begin(), idx.end());
            first20.assign((size_t)nAllelesOf[e], 0); kickedCount.assign((size_t)nAllelesOf[e], 0);
            for(int i = 0; i < prm->first20_n; i++) first20[E[idx[i]].allele]++;
            bool kickedOneOut = false;
            for(int i = 0; i < n; i++) {
                const double first20_prop = (double)first20[E[i].allele] / (double)(prm->filter_first20 != 0);                // sic: the bool (:1593)
                S.considered_alleles++;
                if(first20_prop < prm->first20_min_prop) {
                    ignoreAllele[(size_t)aOff[e] + E[i].allele] = 1; kicked[E[i].read]++; kickedCount[E[i].allele]++; kickedOneOut = true; S.removed_alleles++;
                }
            }
            for(int i = 0; i < n; i++) if(kickedCount[E[i].allele] >= 2) kickedRobust[E[i].read]++;                            // :1628-1641
            S.considered_positions++;
            if(kickedOneOut) S.positions_with_removed_alleles++;
        }
        for(int r = 0; r < nReads; r++) {
            if(kicked[r] > prm->first20_limit_per_read) S.reads_kicked_out++;
            if(kickedRobust[r] > prm->first20_limit_per_read) { S.reads_kicked_out_robust++; ignoreRead[r] = 1; }                // ignore_readIDs, :1686-1690
        }
    }
    // ---- high-coverage allele filter and (long reads) the strand filter, :1722-1862; both look at the counts taken before either
    if(prm->long_read_strand_filter && nReads > 0 && !pos->read_reverse) return HLALA_E_ARG;
    {
        std::vector<int> cnt, rev; std::vector<std::pair<std::string, int>> byName;
        for(int e = 0; e <= maxExon; e++) {
            const int b0 = bOff[e], n = bOff[(size_t)e + 1] - b0;
            if(n == 0) continue;
            cnt.assign((size_t)nAllelesOf[e], 0);
            int count_position = 0;
            rev.assign((size_t)nAllelesOf[e], 0);
            for(int i = 0; i < n; i++) {
                const Entry& E = ent[(size_t)b0 + i]; if(ignoreRead[E.read] || ignoreAllele[(size_t)aOff[e] + E.allele]) continue;
                cnt[E.allele]++; count_position++;
                if(prm->long_read_strand_filter && pos->read_reverse[2 * E.read + (pos->pos_mate[E.pos_index] == 2 ? 1 : 0)]) rev[E.allele]++;
            }
            if(count_position == 0) continue;
            std::vector<hlala_host::AlleleTally>* tal = nullptr;
            if(tallies) {
                // what the reports print per allele of a position (perPosition_allele_counts / _byStrand / _strand1, :1731-1786): the counts of this stage
                if((int)tallies->size() <= e) tallies->resize((size_t)e + 1);
                tal = &(*tallies)[e]; tal->assign((size_t)nAllelesOf[e], hlala_host::AlleleTally());
                for(int i = 0; i < n; i++) {
                    const Entry& E = ent[(size_t)b0 + i]; if(ignoreRead[E.read] || ignoreAllele[(size_t)aOff[e] + E.allele]) continue;
                    hlala_host::AlleleTally& t = (*tal)[E.allele]; const int j = E.pos_index;
                    if(t.count == 0) t.allele.assign((const char*)pos->geno_chars + pos->geno_off[j], (size_t)(pos->geno_off[j + 1] - pos->geno_off[j]));
                    t.count++;
                    if(pos->read_reverse && pos->read_reverse[2 * E.read + (pos->pos_mate[j] == 2 ? 1 : 0)]) t.reverse++;
                    if(pos->pos_mate[j] == 1) t.from_first++;
                }
            }
            if(count_position >= prm->high_coverage_min_coverage) {
                S.high_coverage_positions++;
                for(int a = 0; a < nAllelesOf[e]; a++) {
                    if(cnt[a] == 0) continue;
                    const double aF = (double)cnt[a] / (double)count_position;
                    if((aF < prm->high_coverage_min_freq) && prm->high_coverage_filter) { ignoreAllele[(size_t)aOff[e] + a] = 1; S.high_coverage_removed_alleles += cnt[a]; }
                    else if(tal) (*tal)[a].post_filtering = cnt[a];                                    // perPosition_allele_counts_postFiltering, :1821
                }
            }
            if(prm->long_read_strand_filter) {
                // alleles in std::map<std::string> order: the reference's position counter depends on it (:1857-1860)
                byName.clear();
                for(int i = 0; i < n; i++) { const Entry& E = ent[(size_t)b0 + i]; if(cnt[E.allele] > 0) { const int j = E.pos_index; byName.emplace_back(std::string((const char*)pos->geno_chars + pos->geno_off[j], (size_t)(pos->geno_off[j + 1] - pos->geno_off[j])), E.allele); } }
                std::sort(byName.begin(), byName.end()); byName.erase(std::unique(byName.begin(), byName.end()), byName.end());
                bool kickedOne = false;
                for(const auto& an : byName) {
                    const int a = an.second, total = cnt[a], minStrand = std::min(rev[a], total - rev[a]);
                    const double minStrandFreq = (double)minStrand / (double)total;
                    if(total >= prm->strand_min_allele_coverage) {
                        S.strand_alleles_enough_coverage++;
                        if(minStrandFreq < prm->strand_min_freq) { ignoreAllele[(size_t)aOff[e] + a] = 1; S.strand_removed_alleles++; kickedOne = true; }
                    }
                    if(kickedOne) S.strand_positions_with_removed++;
                }
            }
        }
    }
    // ---- the use test of the likelihood loop, :2102-2120
    for(int r = 0; r < nReads; r++) {
        if(read_ignored) read_ignored[r] = ignoreRead[r];
        for(int j = pos->pos_off[r]; j < pos->pos_off[r + 1]; j++) {
            const bool use = mapqOK[j] && !ignoreAllele[(size_t)aOff[pos->pos_exon[j]] + alleleOf[j]] && !ignoreRead[r];
            pos_use[j] = use ? 1 : 0;
            if(use) S.bases_used++;
        }
    }
    if(stats) *stats = S;
    return HLALA_OK;
}
