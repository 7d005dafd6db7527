// hlala_host.hpp -- C++ host-side mirror of the reference interface of the hot path, on top of the C ABI
// (include/hlala_gpu.h).  Same names and argument meaning as the reference so that call sites read alike:
//   mapper::reads::verboseSeedChain / verboseSeedChainPair      mapper/reads/verboseSeedChain.h:22-346
//   mapper::reads::oneRead                                       mapper/reads/oneRead.h
//   mapper::aligner::extensionAligner::extendSeedChain           mapper/aligner/extensionAligner.h:30
//   mapper::aligner::extensionAligner::scoreOneAlignment         mapper/aligner/extensionAligner.h:34
//   mapper::processBAM::alignOneReadPair (batched here)          mapper/processBAM.h:87
// Error behaviour: where the reference would `assert` / throw, these throw std::runtime_error with the library's
// error text (the reference aborts the process; a maintainer can keep that with a catch-all + abort()).
// Header-only, needs only a C++11 compiler and libhlala_gpu.so -- no HIP headers.
#pragma once
#include <cstdint>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "../../include/hlala_gpu.h"

namespace hlala {
namespace host {

struct Graph {                          // Graph::readFromFile result, creation order (Graph/Graph.cpp:2329-2559)
    int32_t n_levels = 0;
    std::vector<int32_t> node_level, edge_from, edge_to;
    std::vector<uint8_t> edge_label;
};
struct Contigs {                        // reference FASTA + translation/<id>.txt (processBAM.cpp:85-88, 4389-4457)
    std::vector<int64_t> contig_off{0};
    std::vector<uint8_t> contig_seq;
    std::vector<int32_t> contig_level;
    std::vector<int32_t> contig_seqid;
    void add(int32_t seqid, const std::string& bases, const std::vector<int32_t>& levels)
    {
        contig_seq.insert(contig_seq.end(), bases.begin(), bases.end());
        contig_level.insert(contig_level.end(), levels.begin(), levels.end());
        contig_off.push_back((int64_t)contig_seq.size());
        contig_seqid.push_back(seqid);
    }
};

namespace mapper {
namespace reads {

struct verboseSeedChain {               // field names of the reference class; edges are creation indices, -1 = null
    int sequence_begin = -1, sequence_end = -1;
    bool reverse = false;
    int removed_columns_noGap_restriction = -1;
    std::vector<bool> is_from_BWAseed;
    std::vector<int32_t> graph_aligned_edges;
    std::vector<int32_t> graph_aligned_levels;
    std::string graph_aligned, sequence_aligned;
    double mapQ = 0;
    std::string mapQ_perPosition;
    double log_likelihood = 0;          // scoreOneAlignment of this chain (computed by the same launch)
    int alignment_firstLevel() const { for(int l : graph_aligned_levels) if(l != -1) return l; return -1; }
    int alignment_lastLevel() const { for(size_t i = graph_aligned_levels.size(); i-- > 0;) if(graph_aligned_levels[i] != -1) return graph_aligned_levels[i]; return -1; }
};
struct verboseSeedChainPair { std::string readID; std::pair<verboseSeedChain, verboseSeedChain> chains; double mapQ = -1; };
struct oneRead { std::string name, sequence, quality; };

struct BamRecord {                      // the BamTools::BamAlignment members the path reads
    int32_t contig = 0, Position = 0, reference2level_offset = 0, AS = 0;
    bool IsReverseStrand = false, IsPrimaryAlignment = false;
    std::vector<uint32_t> CigarData;    // BAM encoding len<<4|op
};
struct protoSeeds {                     // mapper/reads/protoSeeds.h: alignments in AS-descending order
    std::string readID;
    std::vector<BamRecord> read1_alignments, read2_alignments;
    std::string read1_QueryBases, read1_Qualities, read2_QueryBases, read2_Qualities;   // of the primaries, alignment orientation
};

}  // namespace reads

namespace aligner {

class extensionAligner {
public:
    // `rng_seed` plays extensionAligner::rng_seeds[0] (public mutable member, extensionAligner.h:38)
    extensionAligner(const Graph& g, const Contigs* contigs, double IS_mean, double IS_sd, uint32_t rng_seed = 0, int max_columns = 384,
                     int device = 0, void* stream = nullptr)
    {
        hlala_graph_desc gd{g.n_levels, (int32_t)g.node_level.size(), (int32_t)g.edge_from.size(), g.node_level.data(), g.edge_from.data(), g.edge_to.data(), g.edge_label.data()};
        hlala_contigs_desc cd{};
        if(contigs) { cd.n_contigs = (int32_t)contigs->contig_seqid.size(); cd.contig_off = contigs->contig_off.data(); cd.contig_seq = contigs->contig_seq.data();
                      cd.contig_level = contigs->contig_level.data(); cd.contig_seqid = contigs->contig_seqid.data(); }
        params_ = hlala_params{IS_mean, IS_sd, rng_seed, 0, max_columns, 0};
        if(hlala_create(&ctx_, device, stream, &gd, contigs ? &cd : nullptr, &params_) != HLALA_OK)
            throw std::runtime_error(std::string("hlala_create: ") + hlala_last_error(nullptr));
    }
    ~extensionAligner() { hlala_destroy(ctx_); }
    extensionAligner(const extensionAligner&) = delete;
    extensionAligner& operator=(const extensionAligner&) = delete;
    hlala_ctx* ctx() const { return ctx_; }
    int max_columns() const { return params_.max_columns; }

    // extendSeedChain (extensionAligner.cpp:186) for many (read, seed chain) couples in one launch; chain i belongs to reads[chain_read[i]]
    std::vector<reads::verboseSeedChain> extendSeedChains(const std::vector<std::string>& sequences, const std::vector<std::string>& qualities,
                                                          const std::vector<int>& chain_read, const std::vector<reads::verboseSeedChain>& seedChains) const
    {
        std::vector<int32_t> read_off{0}, col_off{0}, cread, cbeg, cend, lev, edg; std::vector<uint8_t> bases, quals, crev, g, s;
        for(size_t r = 0; r < sequences.size(); r++) {
            bases.insert(bases.end(), sequences[r].begin(), sequences[r].end());
            const std::string& q = r < qualities.size() ? qualities[r] : std::string(sequences[r].size(), 'I');
            quals.insert(quals.end(), q.begin(), q.end());
            read_off.push_back((int32_t)bases.size());
        }
        for(size_t c = 0; c < seedChains.size(); c++) {
            const reads::verboseSeedChain& sc = seedChains[c];
            cread.push_back(chain_read[c]); cbeg.push_back(sc.sequence_begin); cend.push_back(sc.sequence_end); crev.push_back(sc.reverse ? 1 : 0);
            lev.insert(lev.end(), sc.graph_aligned_levels.begin(), sc.graph_aligned_levels.end());
            edg.insert(edg.end(), sc.graph_aligned_edges.begin(), sc.graph_aligned_edges.end());
            g.insert(g.end(), sc.graph_aligned.begin(), sc.graph_aligned.end());
            s.insert(s.end(), sc.sequence_aligned.begin(), sc.sequence_aligned.end());
            col_off.push_back((int32_t)lev.size());
        }
        hlala_seeds_in in{(int32_t)sequences.size(), read_off.data(), bases.data(), quals.data(), (int32_t)seedChains.size(), cread.data(), cbeg.data(), cend.data(),
                          crev.data(), col_off.data(), lev.data(), edg.data(), g.data(), s.data()};
        hlala_batch* b = nullptr;
        check(hlala_batch_create_from_seeds(ctx_, &in, &b), "hlala_batch_create_from_seeds");
        int rc = hlala_extend_chains(ctx_, b);
        std::vector<reads::verboseSeedChain> out;
        if(rc == HLALA_OK) out = fetch_chains(b, (int)seedChains.size(), &crev);
        hlala_batch_destroy(b);
        check(rc, "hlala_extend_chains");
        return out;
    }
    // single-call form with the reference's exact signature
    reads::verboseSeedChain extendSeedChain(const std::string& sequence, const reads::verboseSeedChain& seedChain) const
    {
        return extendSeedChains({sequence}, {}, {0}, {seedChain}).at(0);
    }
    // scoreOneAlignment (extensionAligner.cpp:52): the log-likelihood is produced by the launch that extended the chain
    double scoreOneAlignment(const reads::verboseSeedChain& alignment, const reads::oneRead&, std::string = "") const { return alignment.log_likelihood; }

    // processBAM::alignOneReadPair over a batch of proto seeds (mapper/processBAM.cpp:3129)
    std::vector<reads::verboseSeedChainPair> alignReadPairs(const std::vector<reads::protoSeeds>& seeds) const
    {
        std::vector<int32_t> read_off{0}, chain_off{0}, read_primary, contig, pos, offs, as, cigar_off{0}; std::vector<uint8_t> bases, quals, rev; std::vector<uint32_t> cigar;
        for(const reads::protoSeeds& ps : seeds)
            for(int m = 0; m < 2; m++) {
                const std::vector<reads::BamRecord>& al = m ? ps.read2_alignments : ps.read1_alignments;
                const std::string& qb = m ? ps.read2_QueryBases : ps.read1_QueryBases; const std::string& ql = m ? ps.read2_Qualities : ps.read1_Qualities;
                bases.insert(bases.end(), qb.begin(), qb.end()); quals.insert(quals.end(), ql.begin(), ql.end()); read_off.push_back((int32_t)bases.size());
                int prim = -1;
                for(const reads::BamRecord& a : al) {
                    if(a.IsPrimaryAlignment) prim = (int)contig.size();
                    contig.push_back(a.contig); pos.push_back(a.Position); offs.push_back(a.reference2level_offset); as.push_back(a.AS); rev.push_back(a.IsReverseStrand ? 1 : 0);
                    cigar.insert(cigar.end(), a.CigarData.begin(), a.CigarData.end()); cigar_off.push_back((int32_t)cigar.size());
                }
                if(prim < 0) throw std::runtime_error("protoSeeds without a primary alignment (protoSeeds.cpp:255-330 asserts)");
                read_primary.push_back(prim); chain_off.push_back((int32_t)contig.size());
            }
        hlala_batch_in in{(int32_t)seeds.size(), read_off.data(), bases.data(), quals.data(), chain_off.data(), read_primary.data(), (int32_t)contig.size(), contig.data(),
                          pos.data(), offs.data(), as.data(), rev.data(), cigar_off.data(), cigar.data()};
        hlala_batch* b = nullptr;
        check(hlala_batch_create(ctx_, &in, &b), "hlala_batch_create");
        int rc = hlala_align_batch(ctx_, b);
        std::vector<reads::verboseSeedChainPair> out;
        if(rc == HLALA_OK) {
            int n = (int)seeds.size(), st = params_.max_columns; size_t n2 = 2 * (size_t)n;
            std::vector<int32_t> status(n), best(n2), ncomb(n), ncols(n2), lev(n2 * st), edg(n2 * st); std::vector<double> ll(n), mq(n), mmq(n2);
            std::vector<uint8_t> sv(n), g(n2 * st), s(n2 * st), fs(n2 * st), pq(n2 * st);
            hlala_pairs_out po{status.data(), best.data(), ncomb.data(), ll.data(), mq.data(), mmq.data(), sv.data(), ncols.data(), lev.data(), edg.data(), g.data(), s.data(), fs.data(), pq.data()};
            rc = hlala_batch_get_pairs(ctx_, b, &po);
            for(int p = 0; rc == HLALA_OK && p < n; p++) {
                if(status[p] != 0) { hlala_batch_destroy(b); throw std::runtime_error("alignOneReadPair: a chain of pair " + seeds[p].readID + " exceeded a device capacity"); }
                reads::verboseSeedChainPair vp; vp.readID = seeds[p].readID; vp.mapQ = mq[p];
                for(int m = 0; m < 2; m++) {
                    reads::verboseSeedChain& c = m ? vp.chains.second : vp.chains.first; size_t r = 2 * (size_t)p + m, o = r * st; int k = ncols[r];
                    c.graph_aligned_levels.assign(lev.begin() + o, lev.begin() + o + k); c.graph_aligned_edges.assign(edg.begin() + o, edg.begin() + o + k);
                    c.graph_aligned.assign(g.begin() + o, g.begin() + o + k); c.sequence_aligned.assign(s.begin() + o, s.begin() + o + k);
                    c.mapQ_perPosition.assign(pq.begin() + o, pq.begin() + o + k); c.is_from_BWAseed.assign(fs.begin() + o, fs.begin() + o + k);
                    c.mapQ = mmq[r]; c.reverse = rev[best[r]] != 0; c.sequence_begin = 0; c.sequence_end = read_off[r + 1] - read_off[r] - 1;
                }
                out.push_back(vp);
            }
        }
        hlala_batch_destroy(b);
        check(rc, "hlala_align_batch");
        return out;
    }

private:
    void check(int rc, const char* what) const { if(rc != HLALA_OK) throw std::runtime_error(std::string(what) + ": " + hlala_last_error(ctx_)); }
    std::vector<reads::verboseSeedChain> fetch_chains(hlala_batch* b, int n, const std::vector<uint8_t>* crev) const
    {
        int st = params_.max_columns;
        std::vector<int32_t> status(n), ncols(n), beg(n), end(n), rem(n), it(2 * n), sc(2 * n), lev((size_t)n * st), edg((size_t)n * st); std::vector<double> ll(n);
        std::vector<uint8_t> g((size_t)n * st), s((size_t)n * st), fs((size_t)n * st);
        hlala_chains_out co{status.data(), ncols.data(), beg.data(), end.data(), rem.data(), ll.data(), it.data(), sc.data(), lev.data(), edg.data(), g.data(), s.data(), fs.data()};
        check(hlala_batch_get_chains(ctx_, b, 1, &co), "hlala_batch_get_chains");
        std::vector<reads::verboseSeedChain> out(n);
        for(int c = 0; c < n; c++) {
            if(status[c] != HLALA_CHAIN_OK) throw std::runtime_error("extendSeedChain: chain exceeded a device capacity or has invalid input (status " + std::to_string(status[c]) + ")");
            size_t o = (size_t)c * st; int k = ncols[c]; reads::verboseSeedChain& v = out[c];
            v.sequence_begin = beg[c]; v.sequence_end = end[c]; v.reverse = crev ? (*crev)[c] != 0 : false; v.log_likelihood = ll[c];
            v.graph_aligned_levels.assign(lev.begin() + o, lev.begin() + o + k); v.graph_aligned_edges.assign(edg.begin() + o, edg.begin() + o + k);
            v.graph_aligned.assign(g.begin() + o, g.begin() + o + k); v.sequence_aligned.assign(s.begin() + o, s.begin() + o + k); v.is_from_BWAseed.assign(fs.begin() + o, fs.begin() + o + k);
        }
        return out;
    }
    hlala_ctx* ctx_ = nullptr;
    hlala_params params_{};
};

}  // namespace aligner
}  // namespace mapper
}  // namespace host
}  // namespace hlala
