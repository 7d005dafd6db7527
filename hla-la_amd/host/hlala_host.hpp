// hlala_host.hpp -- C++ host-side mirror of the reference interface of the hot path, on top of the C ABI
// (include/hlala_gpu.h).  Same names and argument meaning as the reference so that call sites read alike:
//   mapper::reads::verboseSeedChain / verboseSeedChainPair      mapper/reads/verboseSeedChain.h:22-346
//   mapper::reads::oneRead                                       mapper/reads/oneRead.h
//   mapper::aligner::extensionAligner::extendSeedChain           mapper/aligner/extensionAligner.h:30
//   mapper::aligner::extensionAligner::scoreOneAlignment         mapper/aligner/extensionAligner.h:34
//   mapper::processBAM::alignOneReadPair (batched here)          mapper/processBAM.h:87
//   mapper::processBAM (graph directory + BAM -> resident batch) mapper/processBAM.cpp:30-158, 703-864, 1071-1165, 1183-1402
//   hla::HLATyper::HLATypeInference                              hla/HLATyper.cpp:934-2810
// Error behaviour: where the reference would `assert` / throw, these throw std::runtime_error with the library's
// error text (the reference aborts the process; a maintainer can keep that with a catch-all + abort()).
// Header-only, needs only a C++11 compiler and libhlala_gpu.so -- no HIP headers.
#pragma once
#include <dirent.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <cerrno>
#include <chrono>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <thread>
#include <utility>
#include <vector>

#include "../../include/hlala_gpu.h"

namespace hlala {
namespace host {

// a libhlala_gpu.so built from another revision of the header can keep every struct size and still mean something else by a field
inline void check_abi()
{
    if(hlala_abi_version() != HLALA_ABI_VERSION)
        throw std::runtime_error("libhlala_gpu.so implements interface version " + std::to_string(hlala_abi_version()) + ", this program was compiled against version " + std::to_string(HLALA_ABI_VERSION) + " of include/hlala_gpu.h");
}
// joins the threads it holds when it goes out of scope: an exception between the start of a thread and its join() must not destroy a joinable
// std::thread (std::terminate)
struct ThreadJoiner {
    std::vector<std::thread> th;
    template <class F> void start(F&& f) { th.emplace_back(std::forward<F>(f)); }
    void join() { for(std::thread& t : th) if(t.joinable()) t.join(); }
    ~ThreadJoiner() { join(); }
};

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
        check_abi();
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
        std::vector<int64_t> read_off{0}, chain_off{0}, cigar_off{0}; std::vector<int32_t> read_primary, contig, pos, offs, as; std::vector<uint8_t> bases, quals, rev; std::vector<uint32_t> cigar;
        for(const reads::protoSeeds& ps : seeds)
            for(int m = 0; m < 2; m++) {
                const std::vector<reads::BamRecord>& al = m ? ps.read2_alignments : ps.read1_alignments;
                const std::string& qb = m ? ps.read2_QueryBases : ps.read1_QueryBases; const std::string& ql = m ? ps.read2_Qualities : ps.read1_Qualities;
                bases.insert(bases.end(), qb.begin(), qb.end()); quals.insert(quals.end(), ql.begin(), ql.end()); read_off.push_back((int64_t)bases.size());
                int prim = -1;
                for(const reads::BamRecord& a : al) {
                    if(a.IsPrimaryAlignment) prim = (int)contig.size();
                    contig.push_back(a.contig); pos.push_back(a.Position); offs.push_back(a.reference2level_offset); as.push_back(a.AS); rev.push_back(a.IsReverseStrand ? 1 : 0);
                    cigar.insert(cigar.end(), a.CigarData.begin(), a.CigarData.end()); cigar_off.push_back((int64_t)cigar.size());
                }
                if(prim < 0) throw std::runtime_error("protoSeeds without a primary alignment (protoSeeds.cpp:255-330 asserts)");
                read_primary.push_back(prim); chain_off.push_back((int64_t)contig.size());
            }
        hlala_batch_in in{(int32_t)seeds.size(), read_off.data(), bases.data(), quals.data(), chain_off.data(), read_primary.data(), (int32_t)contig.size(), contig.data(),
                          pos.data(), offs.data(), as.data(), rev.data(), cigar_off.data(), cigar.data()};
        hlala_batch* b = nullptr;
        check(hlala_batch_create(ctx_, &in, &b), "hlala_batch_create");
        int rc = hlala_align_batch(ctx_, b);
        std::vector<reads::verboseSeedChainPair> out;
        if(rc == HLALA_OK) {
            // per-pair scalars from hlala_batch_get_pairs (no column pointers), the columns without padding from hlala_batch_get_pairs_packed
            int n = (int)seeds.size(); size_t n2 = 2 * (size_t)n;
            std::vector<int32_t> status(n), best(n2), ncomb(n); std::vector<double> ll(n), mq(n), mmq(n2); std::vector<uint8_t> sv(n);
            hlala_pairs_out po{status.data(), best.data(), ncomb.data(), ll.data(), mq.data(), mmq.data(), sv.data(), nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
            rc = hlala_batch_get_pairs(ctx_, b, &po);
            std::vector<int64_t> off(n2 + 1); std::vector<int32_t> lev, edg; std::vector<uint8_t> g, s, fs, pq;
            if(rc == HLALA_OK) {
                hlala_pairs_packed_out pk{}; pk.col_off = off.data(); pk.cap_cols = 0;
                rc = hlala_batch_get_pairs_packed(ctx_, b, &pk);                          // sizing call
                if(rc == HLALA_E_CAPACITY || rc == HLALA_OK) {
                    const size_t T = (size_t)pk.n_cols_total;
                    lev.resize(T + 1); edg.resize(T + 1); g.resize(T + 1); s.resize(T + 1); fs.resize(T + 1); pq.resize(T + 1);
                    pk.cap_cols = pk.n_cols_total; pk.col_level = lev.data(); pk.col_edge = edg.data(); pk.col_gchar = g.data(); pk.col_schar = s.data(); pk.col_fromseed = fs.data(); pk.col_mapq = pq.data();
                    rc = hlala_batch_get_pairs_packed(ctx_, b, &pk);
                }
            }
            for(int p = 0; rc == HLALA_OK && p < n; p++) {
                if(status[p] != 0) { hlala_batch_destroy(b); throw std::runtime_error("alignOneReadPair: a chain of pair " + seeds[p].readID + " exceeded a device capacity"); }
                reads::verboseSeedChainPair vp; vp.readID = seeds[p].readID; vp.mapQ = mq[p];
                for(int m = 0; m < 2; m++) {
                    reads::verboseSeedChain& c = m ? vp.chains.second : vp.chains.first; size_t r = 2 * (size_t)p + m, o = (size_t)off[r]; size_t k = (size_t)(off[r + 1] - off[r]);
                    c.graph_aligned_levels.assign(lev.begin() + o, lev.begin() + o + k); c.graph_aligned_edges.assign(edg.begin() + o, edg.begin() + o + k);
                    c.graph_aligned.assign(g.begin() + o, g.begin() + o + k); c.sequence_aligned.assign(s.begin() + o, s.begin() + o + k);
                    c.mapQ_perPosition.assign(pq.begin() + o, pq.begin() + o + k); c.is_from_BWAseed.assign(fs.begin() + o, fs.begin() + o + k);
                    c.mapQ = mmq[r]; c.reverse = rev[best[r]] != 0; c.sequence_begin = 0; c.sequence_end = (int)(read_off[r + 1] - read_off[r]) - 1;
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

// mapper::processBAM as far as the hot path needs it: the graph directory (serializedGRAPH cache or PRG/graph.txt, sequences.txt,
// reference FASTA, translation files) becomes one context per GPU; a BAM file becomes the seeds of the whole sample (64-bit offsets, decoded on
// `threads` host threads), which go through the GPUs in batches of at most `batchPairs` units (the reference walks 10 000 read IDs at a time on
// its threads, mapper/processBAM.cpp:1794, 2391-2483; BASELINE config 3 is ~10 M pairs).  Batch bi runs on device bi % #devices and keeps the
// absolute numbering of its chains, so every DP draws the random seed it draws in a one-batch, one-GPU run: results do not depend on the
// batch size or on the number of devices.
// What a process reads of a graph directory ONCE, however many samples it aligns (BASELINE config 4: eight samples side by side, one per GPU): the graph,
// the contigs with their translation tables, the reference intervals the BAM decoder keeps.  Read-only after construction; every processBAM of the process
// holds a reference.
class GraphDirectory {
public:
    // Returns once the reference intervals are known (sequences.txt + the reference sequences it names: what the BAM decoder needs); the graph and the
    // translation tables -- the bulk of the directory -- are read on two threads behind it.  graph() / contigs() wait for them.
    GraphDirectory(const std::string& graphDir, bool extendedReferenceGenome) : dir(graphDir), extended(extendedReferenceGenome)
    {
        const auto t0 = std::chrono::steady_clock::now();
        if(hlala_contigs_open_dir(graphDir.c_str(), extendedReferenceGenome ? 1 : 0, &contigs_) != HLALA_OK) throw std::runtime_error(std::string("contigs: ") + hlala_loader_last_error());
        intervals_.resize((size_t)hlala_contigs_file_intervals(contigs_, nullptr, 0));
        hlala_contigs_file_intervals(contigs_, intervals_.data(), (int32_t)intervals_.size());
        intervals_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        // `--action prepareGraph` leaves the flattened arrays in <graphDir>/serializedGRAPH; a file of that name written by the reference
        // binary (a Boost archive) is not ours and the text graph is parsed instead.  The graph and the translation tables (tens of millions of lines) are
        // independent files: read side by side.
        try {
            loaders_.start([this, graphDir, t0]() {
                try {
                    if(hlala_graph_cache_load((graphDir + "/serializedGRAPH").c_str(), &graph_) != HLALA_OK) {
                        graph_ = nullptr;
                        if(hlala_graph_load_text((graphDir + "/PRG/graph.txt").c_str(), &graph_) != HLALA_OK) gErr_ = std::string("graph.txt: ") + hlala_loader_last_error();
                    }
                } catch(const std::exception& e) { gErr_ = e.what(); }
                graph_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            });
            loaders_.start([this, t0]() {
                try { if(hlala_contigs_load_translations(contigs_) != HLALA_OK) cErr_ = std::string("contigs: ") + hlala_loader_last_error(); }
                catch(const std::exception& e) { cErr_ = e.what(); }
                contigs_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            });
        } catch(...) { loaders_.join(); hlala_contigs_file_free(contigs_); if(graph_) hlala_graph_file_free(graph_); throw; }
    }
    ~GraphDirectory() { loaders_.join(); hlala_contigs_file_free(contigs_); if(graph_) hlala_graph_file_free(graph_); }
    GraphDirectory(const GraphDirectory&) = delete;
    GraphDirectory& operator=(const GraphDirectory&) = delete;
    // the graph and the complete contigs (several threads may ask)
    void wait()
    {
        std::lock_guard<std::mutex> g(mu_);
        loaders_.join();
        if(!gErr_.empty() || !cErr_.empty()) throw std::runtime_error(gErr_.empty() ? cErr_ : gErr_);
    }
    hlala_graph_file* graph() { wait(); return graph_; }
    hlala_contigs_file* contigs() { wait(); return contigs_; }
    const std::vector<hlala_bam_interval>& intervals() const { return intervals_; }
    const std::string dir; const bool extended;
    double intervals_seconds = 0, graph_seconds = 0, contigs_seconds = 0;      // since the constructor began: intervals known, graph read, translation tables read
private:
    hlala_graph_file* graph_ = nullptr; hlala_contigs_file* contigs_ = nullptr; std::vector<hlala_bam_interval> intervals_;
    std::string gErr_, cErr_; std::mutex mu_; ThreadJoiner loaders_;
};

class processBAM {
public:
    processBAM(const std::string& graphDir, bool extendedReferenceGenome, int max_columns = 384, uint32_t rng_seed = 0, int device = 0)
        : processBAM(graphDir, extendedReferenceGenome, max_columns, rng_seed, std::vector<int>(1, device), 0) {}
    processBAM(const std::string& graphDir, bool extendedReferenceGenome, int max_columns, uint32_t rng_seed, const std::vector<int>& devices, int threads)
        : processBAM(std::make_shared<GraphDirectory>(graphDir, extendedReferenceGenome), max_columns, rng_seed, devices, threads) {}
    // a graph directory the process has read already (several samples per process share one)
    processBAM(const std::shared_ptr<GraphDirectory>& gdir, int max_columns, uint32_t rng_seed, const std::vector<int>& devices, int threads)
        : gdir_(gdir), graphDir_(gdir->dir), extended_(gdir->extended), max_columns_(max_columns), rng_seed_(rng_seed), devices_(devices.empty() ? std::vector<int>(1, 0) : devices), threads_(threads),
          intervals_(gdir->intervals()) {}
    ~processBAM() { for(hlala_batch* b : live_) if(b) hlala_batch_destroy(b); if(comm_) hlala_comm_destroy(comm_); if(owns_ctx_) for(hlala_ctx* c : ctxs_) if(c) hlala_destroy(c); if(seeds_) hlala_seed_batch_free(seeds_); }
    processBAM(const processBAM&) = delete;
    processBAM& operator=(const processBAM&) = delete;

    // initBAM + extractSeeds2 + estimateInsertSize (mapper/processBAM.cpp:1183-1402, 703-864, 1071-1165): seeds of all complete units, insert
    // size from the first 4000 of them (:1075) on the first device, then every context gets that insert size.  batchPairs = 0: one batch.
    // Round 6, several samples taking turns on a device (HLA-LA.cpp: SampleSchedule): the contexts belong to the CALL, not to the sample -- created once per device (flatten
    // + upload + 12 GB of DP slabs: 2 s, and a dozen device-wide synchronisations that the sample before would feel in its kernels), their pools of device blocks warm from
    // the sample before.  Such a sample opens its BAM with borrowed = true (decode only: nothing here touches a GPU while another sample may be using the contexts) and
    // calls use_contexts() when its turn at the device has come: insert-size estimate (:1071-1165) on the first context, insert size and cleared read counters on all.
    void use_contexts(const std::vector<hlala_ctx*>& ctxs)
    {
        if(ctxs.size() != devices_.size()) throw std::runtime_error("use_contexts: one context per listed device");
        ctxs_ = ctxs; owns_ctx_ = false;
        std::vector<int32_t> scratch((size_t)(n_levels > 1 ? n_levels - 1 : 1));
        for(hlala_ctx* c : ctxs_) if(hlala_get_coverage(c, scratch.data(), 1) != HLALA_OK) throw std::runtime_error(std::string("hlala_get_coverage: ") + hlala_last_error(c));      // (the sample before has read its counters)
        if(ctxs_.size() > 1 && !comm_ && hlala_comm_create(ctxs_.data(), (int)ctxs_.size(), &comm_) != HLALA_OK) throw std::runtime_error(std::string("hlala_comm_create: ") + hlala_comm_last_error(nullptr));
        finish_open();
    }
    void openBAM(const std::string& BAM, bool longReads = false, int32_t batchPairs = 0, bool borrowed = false)
    {
        // the BAM is decoded (all host threads) while the contexts are created (one thread per device: each flattens and uploads the graph)
        const auto t0 = std::chrono::steady_clock::now();
        std::string bamErr;
        check_abi();
        // (the decoder thread and the per-device threads are joined by `tdec` / `th` on every way out of this scope, exceptions included;
        //  everything they capture is declared before them)
        hlala_graph_desc gd; hlala_contigs_desc cd;
        hlala_params pr{200.0, 35.0, rng_seed_, longReads ? 1 : 0, max_columns_, 0};
        if(!borrowed) ctxs_.assign(devices_.size(), nullptr);
        std::vector<std::string> errs(devices_.size());
        ThreadJoiner tdec, th;
        // (the decoder needs the reference intervals only: it starts while the graph directory may still be reading its graph and translation tables)
        tdec.start([&]() {
            // (the bases stay 4-bit packed as the BAM records hold them: a copy instead of an unpacking pass here, half the bytes to upload, unpacked on the device)
            const int32_t seedFlags = std::getenv("HLALA_SEEDS_ASCII") ? 0 : HLALA_SEEDS_PACKED;          // (HLALA_SEEDS_ASCII=1: the decoder unpacks the bases on the host, as before round 4 -- for A/B runs)
            try { if(hlala_bam_extract_seeds_opt(BAM.c_str(), (int32_t)intervals_.size(), intervals_.data(), longReads ? 1 : 0, threads_, seedFlags, &seeds_) != HLALA_OK) bamErr = std::string("BAM: ") + hlala_bam_last_error(); }
            catch(const std::exception& e) { bamErr = e.what(); }
            decode_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        });
        graph_ = gdir_->graph(); contigs_ = gdir_->contigs();                       // (waits for the directory; throws what its readers reported: tdec is joined on the way out)
        directory_wait_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        hlala_graph_file_desc(graph_, &gd);
        if(hlala_contigs_file_desc(contigs_, &cd) != HLALA_OK) throw std::runtime_error(std::string("contigs: ") + hlala_loader_last_error());
        n_levels = gd.n_levels;
        for(size_t d = 0; d < devices_.size() && !borrowed; d++) th.start([&, d]() { if(hlala_create(&ctxs_[d], devices_[d], nullptr, &gd, &cd, &pr) != HLALA_OK) errs[d] = std::string("hlala_create: ") + hlala_last_error(nullptr); });
        th.join();
        context_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        tdec.join();
        if(!bamErr.empty()) throw std::runtime_error(bamErr);
        for(const std::string& e : errs) if(!e.empty()) throw std::runtime_error(e);
        // several devices: the merge steps of the sample's results that live on the devices (the per-level read counters; the per-pair records for callers that want
        // them) go over RCCL between the contexts (include/hlala_gpu.h: hlala_comm_*; contexts that share a device are served by copies)
        if(!borrowed && ctxs_.size() > 1 && hlala_comm_create(ctxs_.data(), (int)ctxs_.size(), &comm_) != HLALA_OK) throw std::runtime_error(std::string("hlala_comm_create: ") + hlala_comm_last_error(nullptr));
        hlala_seed_batch_timing(seeds_, decode_phase_seconds, &decode_threads);
        n_units = hlala_seed_batch_units(seeds_); longReadsMode = longReads; BAM_ = BAM;
        batchPairs_ = batchPairs > 0 ? batchPairs : (n_units > 0 ? (n_units > 0x7FFFFFFF ? 0x7FFFFFFF : (int32_t)n_units) : 1);
        live_.assign((size_t)n_batches(), nullptr); aligned_.assign((size_t)n_batches(), 0);
        if(!borrowed) finish_open();
    }
    // what openBAM does on the contexts once they are this sample's: page-locking mode, insert size
    void finish_open()
    {
        const bool longReads = longReadsMode; const std::string& BAM = BAM_;
        // page-locked: batch uploads are plain DMA (a refusal only costs speed).  Window by window, when a window is handed out (filled, then locked): the 0.2 s that
        // locking the whole sample took lay between the decode and the first batch; HLALA_PIN_WHOLE=1 locks everything here as before round 4
        (void)hlala_seed_batch_pin(seeds_, std::getenv("HLALA_PIN_WHOLE") ? 1 : 2);
        if(!longReads && n_units == 0) throw std::runtime_error("estimateInsertSize: no complete read pair in " + BAM);
        if(!longReads) {                                                              // insert size from this sample
            hlala_batch_in first; window(0, n_units < 4000 ? (int32_t)n_units : 4000, first);
            hlala_insert_size_out is; int rc = hlala_estimate_insert_size(ctxs_[0], &first, &is);
            if(rc) throw std::runtime_error(std::string("estimateInsertSize: ") + hlala_last_error(ctxs_[0]));
            IS_mean = is.mean; IS_sd = is.sd;
            for(hlala_ctx* c : ctxs_) if(hlala_set_insert_size(c, is.mean, is.sd) != HLALA_OK) throw std::runtime_error(std::string("hlala_set_insert_size: ") + hlala_last_error(c));
        }
    }
    // seconds the decoder has spent laying the sample out so far: sizes and offsets at decode time, then the fill of every window handed out since
    // (hlala_seed_batch_window fills what it hands out, beside the GPU's work on the batch before)
    double layout_seconds() const { double s[6] = {0, 0, 0, 0, 0, 0}; int32_t t = 0; if(seeds_) hlala_seed_batch_timing(seeds_, s, &t); return s[5]; }
    int32_t n_batches() const { return n_units == 0 ? 0 : (int32_t)((n_units + batchPairs_ - 1) / batchPairs_); }
    int64_t batch_first_unit(int32_t bi) const { return (int64_t)bi * batchPairs_; }
    int32_t batch_units(int32_t bi) const { int64_t a = (int64_t)bi * batchPairs_, z = a + batchPairs_; return (int32_t)((z > n_units ? n_units : z) - a); }
    int n_devices() const { return (int)ctxs_.size(); }
    int batch_device(int32_t bi) const { return (int)(bi % (int32_t)ctxs_.size()); }
    hlala_ctx* batch_ctx(int32_t bi) const { return ctxs_[(size_t)batch_device(bi)]; }
    // batch bi resident on its GPU: uploaded and, if `align`, run through alignOneReadPair / alignOneLongRead (:3129 / :3618).  Batches of
    // different devices may be acquired from different host threads (one thread per device).
    // (A batch acquired without `align` is only uploaded -- its inputs: hlala_batch_create allocates no outputs -- and aligned by a later acquire(bi, true).)
    hlala_batch* acquire(int32_t bi, bool align)
    {
        hlala_ctx* c = batch_ctx(bi);
        hlala_batch* b = live_.at((size_t)bi);
        int rc = HLALA_OK;
        if(!b) {
            hlala_batch_in in; window(batch_first_unit(bi), batch_units(bi), in);            // (fills the window's units if nobody asked for them yet)
            rc = longReadsMode ? hlala_batch_create_unpaired(c, &in, &b) : hlala_batch_create(c, &in, &b);       // (chain_off[0] of the window is the batch's first absolute chain number)
            if(rc != HLALA_OK) { std::string e = hlala_last_error(c); if(b) hlala_batch_destroy(b); throw std::runtime_error("alignReads: " + e); }
            live_[(size_t)bi] = b; aligned_[(size_t)bi] = 0;
        }
        if(align && !aligned_[(size_t)bi]) {
            rc = hlala_align_batch(c, b);
            if(rc != HLALA_OK) { std::string e = hlala_last_error(c); hlala_batch_destroy(b); live_[(size_t)bi] = nullptr; throw std::runtime_error("alignReads: " + e); }
            aligned_[(size_t)bi] = 1;
        }
        return b;
    }
    void release(int32_t bi) { if(live_.at((size_t)bi)) { hlala_batch_destroy(live_[(size_t)bi]); live_[(size_t)bi] = nullptr; aligned_[(size_t)bi] = 0; } }

    // extractSeeds2 + estimateInsertSize + alignReads_postSeedExtraction (mapper/processBAM.cpp:703-864, 1071-1165, 2391-2483), one batch
    void alignReads(const std::string& BAM, bool longReads = false) { openBAM(BAM, longReads, 0); if(n_units > 0) acquire(0, true); }
    hlala_ctx* ctx() const { return ctxs_.empty() ? nullptr : ctxs_[0]; }
    // hlala_set_tail_pool on every context (1: off).  Memory: k + 1 batches' outputs live per device.
    void set_tail_pool(int k) { tail_pool_ = k < 1 ? 1 : k; for(hlala_ctx* c : ctxs_) if(c && hlala_set_tail_pool(c, tail_pool_) != HLALA_OK) throw std::runtime_error(std::string("hlala_set_tail_pool: ") + hlala_last_error(c)); }
    int tail_pool() const { return tail_pool_; }
    bool uses_rccl() const { return comm_ && hlala_comm_uses_rccl(comm_); }          // the contexts sit on different GPUs and exchange over RCCL
    hlala_batch* batch() const { return live_.empty() ? nullptr : live_[0]; }
    const char* readID(int64_t unit) const { return hlala_seed_batch_name(seeds_, unit); }
    // bases_per_level summed over the devices (reads_per_level.txt, processBAM.cpp:1902-1913: the per-thread counters are added up, :1866-1887)
    std::vector<int32_t> coverage() const
    {
        std::vector<int32_t> cov((size_t)(n_levels > 1 ? n_levels - 1 : 1), 0), one(cov.size());
        if(comm_) { if(hlala_reduce_coverage(comm_, cov.data(), 0) != HLALA_OK) throw std::runtime_error(std::string("hlala_reduce_coverage: ") + hlala_comm_last_error(comm_)); return cov; }
        for(hlala_ctx* c : ctxs_) { if(hlala_get_coverage(c, one.data(), 0) != HLALA_OK) throw std::runtime_error(std::string("hlala_get_coverage: ") + hlala_last_error(c)); for(size_t i = 0; i < cov.size(); i++) cov[i] += one[i]; }
        return cov;
    }
    double IS_mean = 200.0, IS_sd = 35.0;
    int64_t n_units = 0; int32_t n_levels = 0; bool longReadsMode = false;
    double decode_seconds = 0, decode_phase_seconds[6] = {0, 0, 0, 0, 0, 0}; int32_t decode_threads = 0;
    double context_seconds = 0;       // until the contexts exist (beside the decode): the wait for the graph directory + their creation
    double directory_wait_seconds = 0;      // ... of which waiting for the graph directory's readers

private:
    void window(int64_t u0, int32_t n, hlala_batch_in& in) const
    {
        if(hlala_seed_batch_window(seeds_, u0, n, &in) != HLALA_OK) throw std::runtime_error(std::string("alignReads: ") + hlala_bam_last_error());
    }
    std::shared_ptr<GraphDirectory> gdir_;
    std::string graphDir_; bool extended_; int max_columns_; uint32_t rng_seed_; std::vector<int> devices_; int threads_;
    hlala_graph_file* graph_ = nullptr; hlala_contigs_file* contigs_ = nullptr; const std::vector<hlala_bam_interval>& intervals_;      // owned by gdir_
    hlala_seed_batch* seeds_ = nullptr; std::vector<hlala_ctx*> ctxs_; hlala_comm* comm_ = nullptr; int tail_pool_ = 1; bool owns_ctx_ = true; std::string BAM_;
    int32_t batchPairs_ = 1; std::vector<hlala_batch*> live_; std::vector<char> aligned_;       // aligned_[bi]: hlala_align_batch has been queued for live_[bi]
};
}  // namespace mapper

// Result tables that a GPU call fills completely (millions of doubles per locus): page-aligned, huge pages where the kernel grants them, NOT cleared -- a
// std::vector zero-fills what the download overwrites a moment later, one 4 KB page fault after the other.
template <class T>
struct RawBuf {
    T* p = nullptr; size_t n = 0;
    RawBuf() {}
    ~RawBuf() { std::free(p); }
    RawBuf(const RawBuf&) = delete; RawBuf& operator=(const RawBuf&) = delete;
    void alloc(size_t count)
    {
        std::free(p); p = nullptr; n = count;
        const size_t H = (size_t)2 << 20, bytes = ((count ? count : 1) * sizeof(T) + H - 1) / H * H;
        void* q = nullptr;
        if(posix_memalign(&q, H, bytes) != 0 || !q) throw std::bad_alloc();
        (void)madvise(q, bytes, MADV_HUGEPAGE);
        p = (T*)q; p[0] = T();
    }
    T* data() { return p; } const T* data() const { return p; }
};

namespace hla {

// hla::HLATyper: graph loci, exon files and allele clusters from the graph directory; HLATypeInference runs the per-locus chain on the
// batch a processBAM left on the GPU and writes the reference's result files (hla/HLATyper.cpp:934-2810).
class HLATyper {
public:
    HLATyper(const std::string& graphDir, const std::string& hla_nom_g = "")
    {
        if(hlala_typer_open(graphDir.c_str(), &t_) != HLALA_OK) throw std::runtime_error(std::string("HLATyper: ") + hlala_typer_last_error());
        if(!hla_nom_g.empty() && hlala_typer_load_g_groups(t_, hla_nom_g.c_str()) != HLALA_OK) { std::string e = hlala_typer_last_error(); hlala_typer_close(t_); throw std::runtime_error("HLATyper: " + e); }
    }
    // the typer's view of a graph directory the process has read already (several samples per process: the handle is read-only, each sample has its own
    // parameters and timing)
    struct Borrow {};
    HLATyper(const HLATyper& loaded, Borrow) : t_(loaded.t_), owns_(false) {}
    ~HLATyper() { if(owns_) hlala_typer_close(t_); }
    HLATyper(const HLATyper&) = delete;
    HLATyper& operator=(const HLATyper&) = delete;

    // the reference's defaults (hla/HLATyper.cpp:28-31, 67-79; HLATyper.h:57)
    hlala_filter_params filterParams{1, 20, 0.1, 2, 0.7, 0, 100, 0.2, 0, 100, 0.1};
    double minimumMappingQuality = 0.0, min_bothReads_weightedCharactersOK = 0.0;
    int minAlignmentLength_unpaired = 1000, k_for_kMer_index = 31;

    struct bestGuess { std::string locus, allele1, allele2; double Q1_allele1 = 0, Q1_allele2 = 0; hlala_locus_report_out summary; };
    // wall clock of the phases of the last HLATypeInference: the walk over the batches (alignment + what the typing takes from a resident batch), the summary
    // file, the per-locus chain (filters, likelihoods, all pairs, call, k-mer questions), the result files after the last locus (kmers: 0 since the reads are kept on the device)
    struct Timing { double batches = 0, summary = 0, loci = 0, kmers = 0, files = 0; } timing;

    // Per-batch part of alignReads_postSeedExtraction_andStoreInto + HLATypeInference, then the per-locus chain.  The sample's units go
    // through the GPU batch by batch (processBAM::acquire); what the typing needs of a batch -- includeInHLA, the per-unit alignment
    // statistics and the exon positions of every locus -- is taken while the batch is resident and appended on the host.
    // align_seconds (optional) receives the time spent in alignment proper (the reference's "Speed:" line, processBAM.cpp:1894-1898).
    std::vector<bestGuess> HLATypeInference(mapper::processBAM& pB, const std::string& outputDirectory, const std::vector<std::string>& loci_for_HLAtyping,
                                            double* align_seconds = nullptr, int64_t* chain_errors = nullptr)
    {
        hlala_ctx* c = pB.ctx();                          // the per-locus chain (likelihoods, all pairs, call) runs on the first device
        auto chk = [&](int rc, const char* what) { if(rc != HLALA_OK) throw std::runtime_error(std::string(what) + ": " + hlala_last_error(c)); };
        auto tchk = [&](int rc, const char* what) { if(rc != HLALA_OK) throw std::runtime_error(std::string(what) + ": " + hlala_typer_last_error()); };
        const int nDev = pB.n_devices();
        // interestingLevels -> coverage counters and includeInHLA (mapper/processBAM.cpp:2411-2446)
        std::vector<int32_t> first, last;
        for(int32_t i = 0; i < hlala_typer_n_genes(t_); i++) { const char* nm; int32_t a, z; hlala_typer_gene(t_, i, &nm, &a, &z); first.push_back(a); last.push_back(z); }
        for(int d = 0; d < nDev; d++) { hlala_ctx* cd = pB.batch_ctx(d); if(hlala_set_gene_intervals(cd, (int32_t)first.size(), first.data(), last.data()) != HLALA_OK) throw std::runtime_error(std::string("hlala_set_gene_intervals: ") + hlala_last_error(cd)); }
        const size_t nU = (size_t)pB.n_units;
        std::vector<uint8_t> include(nU + 1);
        tchk(hlala_typer_begin_output(outputDirectory.c_str(), 0.2), "hlala_typer_begin_output");
        std::vector<uint8_t> usValid(nU + 1), usStrands(nU + 1); std::vector<int32_t> usDist(nU + 1), usCols(2 * nU + 1); std::vector<double> usF(2 * nU + 1), usW(2 * nU + 1), usQ(2 * nU + 1);
        if(pB.longReadsMode) filterParams.long_read_strand_filter = 1;
        // ---- loci
        struct Acc {
            std::string locus; hlala_locus* L = nullptr; hlala_locus_info li;
            std::vector<int32_t> read_pair, read_distance, cols, pos_off{0}, pos_exon, pos_level, novel, geno_off{0};
            std::vector<double> wok, fok, rmapq; std::vector<uint8_t> mate, pmapq, geno, qual, rrev; int32_t ok = 0, broken = 0;
        };
        std::vector<Acc> acc(loci_for_HLAtyping.size());
        struct FreeAll { std::vector<Acc>& a; ~FreeAll() { for(Acc& x : a) if(x.L) hlala_locus_free(x.L); } } guard{acc};
        for(size_t i = 0; i < acc.size(); i++) {
            acc[i].locus = loci_for_HLAtyping[i];
            tchk(hlala_typer_locus(t_, acc[i].locus.c_str(), 0, nullptr, &acc[i].L), "hlala_typer_locus");
            hlala_locus_get(acc[i].L, &acc[i].li);
        }
        // ---- batches.  One host thread per device walks the batches of its device (bi % #devices), two in flight: the next batch is uploaded and its
        // alignment queued before the current one is read back (the reader calls wait for their own batch only).  What the typing needs of a batch is
        // kept per batch (exon positions of every locus: a few thousand reads) and appended in batch order afterwards -- the order of one walk over the
        // sample on one device (the reference adds up its per-thread results in thread order, mapper/processBAM.cpp:1866-1887).
        struct LocusPart { std::vector<int32_t> read_pair, read_distance, cols, pos_off, pos_exon, pos_level, novel, geno_off; std::vector<double> wok, fok, rmapq; std::vector<uint8_t> mate, pmapq, geno, qual, rrev; int32_t ok = 0, broken = 0; size_t nR = 0, nP = 0, nC = 0; };
        const int32_t nB = pB.n_batches();
        std::vector<std::vector<LocusPart>> parts((size_t)nB, std::vector<LocusPart>(acc.size()));
        std::vector<double> devAlign((size_t)nDev, 0.0); std::vector<int64_t> devErrors((size_t)nDev, 0); std::vector<std::string> devErr((size_t)nDev);
        for(int d = 0; d < nDev; d++) hlala_kmer_forget_reads(pB.batch_ctx(d));
        const auto tAll = std::chrono::steady_clock::now();
        auto device_walk = [&](int d) {
            try {
                hlala_ctx* cd = pB.batch_ctx(d);
                auto dchk = [&](int rc, const char* what) { if(rc != HLALA_OK) throw std::runtime_error(std::string(what) + ": " + hlala_last_error(cd)); };
                for(int32_t bi = d; bi < nB; bi += nDev) {
                    const size_t u0 = (size_t)pB.batch_first_unit(bi);
                    const auto t0 = std::chrono::steady_clock::now();
                    hlala_batch* b = pB.acquire(bi, true);
                    // two alignments in flight per device -- with a tail pool of k (round 6: hlala_set_tail_pool, the broad / large / in-memory DP classes of k consecutive batches
                    // in one launch per class) k more than the one being read, so that a whole group is queued before its first batch is waited for ...
                    const int32_t ahead = pB.tail_pool() > 1 ? pB.tail_pool() : 1;
                    for(int32_t j = 1; j <= ahead; j++) if(bi + j * nDev < nB) pB.acquire(bi + j * nDev, true);
                    if(bi + (ahead + 1) * nDev < nB) pB.acquire(bi + (ahead + 1) * nDev, false);      // ... and one upload ahead: the window after those is filled and uploaded (inputs only) while the GPU is busy and before
                                                                                // this thread waits for batch bi -- the host's 100 ms per batch off the critical path (bench.py's boundary loop does the same)
                    hlala_batch_stats bs; dchk(hlala_batch_get_stats(cd, b, &bs), "hlala_batch_get_stats");          // (waits for this batch only)
                    devAlign[(size_t)d] += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
                    devErrors[(size_t)d] += bs.n_errors;
                    dchk(hlala_postprocess_pairs(cd, b, include.data() + u0), "hlala_postprocess_pairs");
                    // the reads of the looked-at units stay on the device for the k-mer questions that follow the calls (HLATyper.cpp:999-1027 builds its index of
                    // read k-mers in this walk too): a few per cent of a batch, copied device to device -- no batch is uploaded a second time
                    dchk(hlala_kmer_keep_reads(cd, b, include.data() + u0, nullptr), "hlala_kmer_keep_reads");
                    // read alignment statistics (hla/HLATyper.cpp:1030-1125) and the per-pair quantities of the histogram lines
                    hlala_unit_stats_out usb{usValid.data() + u0, usStrands.data() + u0, usDist.data() + u0, usF.data() + 2 * u0, usW.data() + 2 * u0, usCols.data() + 2 * u0, usQ.data() + 2 * u0};
                    dchk(hlala_unit_alignment_stats(cd, b, &usb), "hlala_unit_alignment_stats");
                    for(size_t li = 0; li < acc.size(); li++) {
                        Acc& A = acc[li]; LocusPart& P = parts[(size_t)bi][li];
                        hlala_locus_desc ld{A.li.level_min, A.li.level_max, A.li.level_to_exon, pB.IS_mean, pB.IS_sd, minimumMappingQuality, min_bothReads_weightedCharactersOK, include.data() + u0, minAlignmentLength_unpaired, 0};
                        hlala_exon_positions_out pos; std::memset(&pos, 0, sizeof(pos));
                        int rc = hlala_exon_positions(cd, b, &ld, &pos);                                             // sizing call
                        if(rc != HLALA_OK && rc != HLALA_E_CAPACITY) dchk(rc, "hlala_exon_positions");
                        const size_t nR = (size_t)pos.n_reads, nP = (size_t)pos.n_pos, nC = (size_t)pos.n_chars;
                        P.nR = nR; P.nP = nP; P.nC = nC;
                        P.read_pair.resize(nR + 1); P.read_distance.resize(nR + 1); P.cols.resize(2 * nR + 1); P.pos_off.resize(nR + 2); P.pos_exon.resize(nP + 1); P.pos_level.resize(nP + 1); P.novel.resize(nP + 1); P.geno_off.resize(nP + 2);
                        P.wok.resize(2 * nR + 1); P.fok.resize(2 * nR + 1); P.rmapq.resize(2 * nR + 1); P.mate.resize(nP + 1); P.pmapq.resize(nP + 1); P.geno.resize(nC + 1); P.qual.resize(nC + 1); P.rrev.resize(2 * nR + 1);
                        pos.cap_reads = (int32_t)nR; pos.cap_pos = (int32_t)nP; pos.cap_chars = (int32_t)nC;
                        pos.read_pair = P.read_pair.data(); pos.read_weighted_ok = P.wok.data(); pos.read_fraction_ok = P.fok.data(); pos.read_distance = P.read_distance.data(); pos.read_cols_nongap = P.cols.data();
                        pos.pos_off = P.pos_off.data(); pos.pos_exon = P.pos_exon.data(); pos.pos_level = P.pos_level.data(); pos.pos_mate = P.mate.data(); pos.pos_mapq = P.pmapq.data(); pos.pos_novel_gap = P.novel.data();
                        pos.geno_off = P.geno_off.data(); pos.geno_chars = P.geno.data(); pos.qual_chars = P.qual.data(); pos.read_reverse = P.rrev.data(); pos.read_mapq = P.rmapq.data();
                        if(nR > 0) dchk(hlala_exon_positions(cd, b, &ld, &pos), "hlala_exon_positions");
                        P.ok = pos.n_pairs_ok; P.broken = pos.n_pairs_broken;
                    }
                    if(nB > 1) pB.release(bi);
                }
            } catch(const std::exception& e) { devErr[(size_t)d] = e.what(); }
        };
        if(nDev == 1) device_walk(0);
        else { std::vector<std::thread> th; for(int d = 0; d < nDev; d++) th.emplace_back(device_walk, d); for(std::thread& t : th) t.join(); }
        for(const std::string& e : devErr) if(!e.empty()) throw std::runtime_error(e);
        double alignS = nDev == 1 ? devAlign[0] : std::chrono::duration<double>(std::chrono::steady_clock::now() - tAll).count();
        auto lap = [](std::chrono::steady_clock::time_point& t) { const auto n = std::chrono::steady_clock::now(); const double d = std::chrono::duration<double>(n - t).count(); t = n; return d; };
        auto tLap = tAll;
        timing = Timing();
        int64_t errors = 0; for(int64_t e : devErrors) errors += e;
        for(int32_t bi = 0; bi < nB; bi++) {
            const size_t u0 = (size_t)pB.batch_first_unit(bi);
            for(size_t li = 0; li < acc.size(); li++) {
                Acc& A = acc[li]; LocusPart& P = parts[(size_t)bi][li];
                const int32_t pBase = A.pos_off.back(), gBase = A.geno_off.back();
                for(size_t r = 0; r < P.nR; r++) { A.read_pair.push_back(P.read_pair[r] + (int32_t)u0); A.read_distance.push_back(P.read_distance[r]); A.pos_off.push_back(P.pos_off[r + 1] + pBase);
                    for(int m = 0; m < 2; m++) { A.cols.push_back(P.cols[2 * r + m]); A.wok.push_back(P.wok[2 * r + m]); A.fok.push_back(P.fok[2 * r + m]); A.rmapq.push_back(P.rmapq[2 * r + m]); A.rrev.push_back(P.rrev[2 * r + m]); } }
                for(size_t j = 0; j < P.nP; j++) { A.pos_exon.push_back(P.pos_exon[j]); A.pos_level.push_back(P.pos_level[j]); A.novel.push_back(P.novel[j]); A.mate.push_back(P.mate[j]); A.pmapq.push_back(P.pmapq[j]); A.geno_off.push_back(P.geno_off[j + 1] + gBase); }
                A.geno.insert(A.geno.end(), P.geno.begin(), P.geno.begin() + P.nC); A.qual.insert(A.qual.end(), P.qual.begin(), P.qual.begin() + P.nC);
                A.ok += P.ok; A.broken += P.broken;
                LocusPart().pos_off.swap(P.pos_off);
            }
        }
        { std::vector<std::vector<LocusPart>>().swap(parts); }
        timing.batches = lap(tLap);
        if(align_seconds) *align_seconds = alignS;
        if(chain_errors) *chain_errors = errors;
        hlala_unit_stats_out us{usValid.data(), usStrands.data(), usDist.data(), usF.data(), usW.data(), usCols.data(), usQ.data()};
        tchk(hlala_typer_write_summary(outputDirectory.c_str(), (int32_t)pB.n_units, pB.longReadsMode ? 1 : 0, include.data(), &us, pB.IS_mean, pB.IS_sd, minAlignmentLength_unpaired), "hlala_typer_write_summary");
        timing.summary = lap(tLap);
        // ---- per locus: filters -> likelihoods -> all pairs -> call
        struct Res { hlala_exon_positions_out pos; RawBuf<double> pairLL, misAvg, misMin, pNorm; RawBuf<int32_t> order; hlala_call_out call; std::vector<char> q[2]; int32_t nq[2], nt[2]; std::vector<uint8_t> present[2]; };
        std::vector<Res> res(acc.size());
        // (the all-pairs table of a locus -- millions of lines for class I -- is written by a thread of its own as soon as the locus is called: beside the typing
        // of the next locus; into the locus' directory under the output directory, see "files" below)
        std::vector<std::string> tmpDir(acc.size());
        for(size_t li = 0; li < acc.size(); li++) tmpDir[li] = outputDirectory + "/.locus_" + std::to_string(li);
        std::vector<std::thread> pairWriters; std::vector<std::string> pairErr(acc.size());
        struct JoinAll { std::vector<std::thread>& t; ~JoinAll() { for(std::thread& x : t) if(x.joinable()) x.join(); } } joinPairWriters{pairWriters};
        double lociClock[6] = {0, 0, 0, 0, 0, 0};      // filters (all loci side by side), buffers, likelihoods + all pairs + call, k-mer lists (HLALA_HOST_DEBUG=1 prints them)
        auto tL = std::chrono::steady_clock::now();
        // the host part of every locus first, the loci side by side: the read / allele filters (hla/HLATyper.cpp:1509-1880; host code by design: libstdc++'s tie order)
        // and what the likelihood kernel reads of a position -- first genotype character, genotype length, first quality (hla/HLATyper.cpp:2080-2277)
        struct Prep { std::vector<uint8_t> use, ignored, g0, q0; std::vector<int32_t> glen; hlala_filter_stats fs; size_t nR = 0, nP = 0; std::string err; };
        std::vector<Prep> prep(acc.size());
        {
            auto prepare = [&](size_t li) {
                try {
                    Acc& A = acc[li]; Res& R = res[li]; Prep& Q = prep[li];
                    const size_t nR = A.read_pair.size(), nP = A.pos_exon.size();
                    Q.nR = nR; Q.nP = nP;
                    // (vectors may be empty: keep the pointers valid)
                    A.read_pair.push_back(0); A.read_distance.push_back(0); A.pos_exon.push_back(0); A.pos_level.push_back(0); A.novel.push_back(0); A.mate.push_back(0); A.pmapq.push_back(0);
                    A.geno.push_back(0); A.qual.push_back(0); for(int m = 0; m < 2; m++) { A.cols.push_back(0); A.wok.push_back(0); A.fok.push_back(0); A.rmapq.push_back(0); A.rrev.push_back(0); }
                    hlala_exon_positions_out& pos = R.pos; std::memset(&pos, 0, sizeof(pos));
                    pos.cap_reads = pos.n_reads = (int32_t)nR; pos.cap_pos = pos.n_pos = (int32_t)nP; pos.cap_chars = pos.n_chars = (int32_t)(A.geno.size() - 1); pos.n_pairs_ok = A.ok; pos.n_pairs_broken = A.broken;
                    pos.read_pair = A.read_pair.data(); pos.read_weighted_ok = A.wok.data(); pos.read_fraction_ok = A.fok.data(); pos.read_distance = A.read_distance.data(); pos.read_cols_nongap = A.cols.data();
                    pos.pos_off = A.pos_off.data(); pos.pos_exon = A.pos_exon.data(); pos.pos_level = A.pos_level.data(); pos.pos_mate = A.mate.data(); pos.pos_mapq = A.pmapq.data(); pos.pos_novel_gap = A.novel.data();
                    pos.geno_off = A.geno_off.data(); pos.geno_chars = A.geno.data(); pos.qual_chars = A.qual.data(); pos.read_reverse = A.rrev.data(); pos.read_mapq = A.rmapq.data();
                    Q.use.assign(nP + 1, 0); Q.ignored.assign(nR + 1, 0);
                    if(hlala_filter_positions(&pos, &filterParams, Q.use.data(), Q.ignored.data(), &Q.fs) != HLALA_OK) throw std::runtime_error("hlala_filter_positions failed");
                    Q.g0.assign(nP + 1, 0); Q.q0.assign(nP + 1, 0); Q.glen.assign(nP + 1, 0);
                    for(size_t j = 0; j < nP; j++) { Q.g0[j] = A.geno[A.geno_off[j]]; Q.q0[j] = A.qual[A.geno_off[j]]; Q.glen[j] = A.geno_off[j + 1] - A.geno_off[j]; }
                } catch(const std::exception& e) { prep[li].err = e.what(); }
            };
            if(acc.size() <= 1) { for(size_t li = 0; li < acc.size(); li++) prepare(li); }
            else { ThreadJoiner th; for(size_t li = 0; li < acc.size(); li++) th.start([&, li]() { prepare(li); }); th.join(); }
            for(const Prep& Q : prep) if(!Q.err.empty()) throw std::runtime_error(Q.err);
        }
        lociClock[0] += lap(tL);
        // (read names: the report looks up the units of the reads at the locus only -- a few thousand of the sample's millions; zero pages of the table that nobody touches stay unmapped)
        struct Free { void operator()(const char** p) const { std::free((void*)p); } };
        std::unique_ptr<const char*[], Free> names((const char**)std::calloc(nU + 1, sizeof(const char*)));
        if(!names) throw std::bad_alloc();
        for(const Acc& A : acc) for(int32_t u : A.read_pair) if(u >= 0 && (size_t)u < nU && !names[(size_t)u]) names[(size_t)u] = pB.readID((int64_t)u);
        // The files of a locus are written by a thread of its own as soon as the locus is called and its k-mer questions are answered -- the all-pairs table (millions
        // of lines for class I), then pile-up, read IDs, column incompatibilities, best guesses -- beside the typing of the next locus: every locus into a directory of
        // its own under the output directory; afterwards, in locus order, the rows it appended to the shared files (best guesses, histogram lines) are appended to the
        // real ones and its own files are moved up -- the bytes of one locus after the other.
        std::vector<bestGuess> out(acc.size()); std::string lociJoined;
        std::vector<std::string> ferr(acc.size());
        auto write_one = [&](size_t li) {
            try {
                Acc& A = acc[li]; Res& R = res[li];
                if(::mkdir(tmpDir[li].c_str(), 0775) != 0 && errno != EEXIST) throw std::runtime_error("cannot create " + tmpDir[li]);
                double covered[2];
                for(int a = 0; a < 2; a++) { int hit = 0; for(int32_t i = 0; i < R.nq[a]; i++) hit += R.present[a][i]; covered[a] = R.nt[a] ? (double)hit / (double)R.nt[a] : -1; }
                hlala_locus_report_in rin; std::memset(&rin, 0, sizeof(rin));
                rin.pos = &R.pos; rin.filter = &filterParams; rin.unit_name_1 = names.get(); rin.unit_name_2 = pB.longReadsMode ? nullptr : names.get(); rin.long_read_mode = pB.longReadsMode ? 1 : 0;
                rin.n_clusters = A.li.n_clusters; rin.pair_ll = R.pairLL.data(); rin.mis_avg = R.misAvg.data(); rin.mis_min = R.misMin.data(); rin.order = R.order.data(); rin.p_normalized = R.pNorm.data(); rin.call = &R.call;
                rin.kmers_covered[0] = covered[0]; rin.kmers_covered[1] = covered[1]; rin.unaccounted_min_coverage = 30; rin.unaccounted_min_fraction = 0.2; rin.pairs_file_done = 1;
                rin.unit_stats = &us; rin.unit_mask = include.data(); rin.n_units = (int32_t)pB.n_units; rin.insert_mean = pB.IS_mean; rin.insert_sd = pB.IS_sd;
                rin.min_mapq = minimumMappingQuality; rin.min_weighted_ok = min_bothReads_weightedCharactersOK;
                bestGuess g; g.locus = A.locus;
                if(hlala_locus_write_files(A.L, &rin, tmpDir[li].c_str(), &g.summary) != HLALA_OK) throw std::runtime_error(std::string("hlala_locus_write_files: ") + hlala_typer_last_error());
                g.allele1 = hlala_locus_cluster_id(A.L, R.call.first_cluster); g.allele2 = hlala_locus_cluster_id(A.L, R.call.second_cluster); g.Q1_allele1 = R.call.first_marginal; g.Q1_allele2 = R.call.second_p;
                out[li] = g;
            } catch(const std::exception& e) { ferr[li] = e.what(); }
        };
        JoinAll joinWritersFirst{pairWriters};       // (declared after everything the writers touch: on the way out of an exception they are joined before any of it goes)
        struct Forget { mapper::processBAM& p; int n; ~Forget() { for(int d = 0; d < n; d++) hlala_kmer_forget_reads(p.batch_ctx(d)); } } forget{pB, nDev};
        for(size_t li = 0; li < acc.size(); li++) {
            Acc& A = acc[li]; Res& R = res[li]; Prep& Q = prep[li];
            const size_t nR = Q.nR, nP = Q.nP;
            tL = std::chrono::steady_clock::now();
            hlala_exon_in xin{A.li.n_clusters, A.li.n_columns, A.li.cluster_seq, (int32_t)nR, A.pos_off.data(), A.pos_exon.data(), Q.g0.data(), Q.glen.data(), Q.q0.data(), Q.use.data()};
            const size_t C = (size_t)A.li.n_clusters, nPairs = C * (C + 1) / 2;
            std::vector<double> marginal(C + 1);
            R.pairLL.alloc(nPairs + 1); R.misAvg.alloc(nPairs + 1); R.misMin.alloc(nPairs + 1); R.pNorm.alloc(nPairs + 1); R.order.alloc(nPairs + 1);
            lociClock[1] += lap(tL);
            // per-read likelihoods -> all pairs -> call, the tables left on the device in between (the per-read table of a class-I locus is 100 MB that nobody on the host reads)
            chk(hlala_type_locus(c, &xin, nullptr, nullptr, R.pairLL.data(), R.misAvg.data(), R.misMin.data(), R.order.data(), R.pNorm.data(), marginal.data(), &R.call), "hlala_type_locus");
            lociClock[2] += lap(tL);
            if(std::getenv("HLALA_HOST_DEBUG")) std::cerr << "host-debug: locus " << A.locus << ": " << C << " clusters, " << nR << " reads, " << nP << " positions\n";
            for(int a = 0; a < 2; a++) {                                              // k-mers of the two called alleles, :2652-2688 ...
                const int32_t cl = a ? R.call.second_cluster : R.call.first_cluster; R.nq[a] = 0; R.nt[a] = 0;
                hlala_locus_cluster_kmers(A.L, cl, k_for_kMer_index, nullptr, 0, &R.nq[a], &R.nt[a]);
                R.q[a].assign((size_t)R.nq[a] * k_for_kMer_index + 1, 0); R.present[a].assign((size_t)R.nq[a] + 1, 0);
                tchk(hlala_locus_cluster_kmers(A.L, cl, k_for_kMer_index, R.q[a].data(), R.nq[a], &R.nq[a], &R.nt[a]), "hlala_locus_cluster_kmers");
                // ... and which of them occur in the reads that went into typing: asked of the reads every device kept while it walked its batches
                for(int d = 0; d < nDev; d++) {
                    hlala_ctx* cd = pB.batch_ctx(d);
                    std::vector<uint8_t> pr((size_t)R.nq[a] + 1);
                    if(hlala_kmer_presence_kept(cd, k_for_kMer_index, R.nq[a], R.q[a].data(), pr.data()) != HLALA_OK) throw std::runtime_error(std::string("hlala_kmer_presence_kept: ") + hlala_last_error(cd));
                    for(int32_t i = 0; i < R.nq[a]; i++) R.present[a][(size_t)i] |= pr[(size_t)i];
                }
            }
            if(::mkdir(tmpDir[li].c_str(), 0775) != 0 && errno != EEXIST) throw std::runtime_error("cannot create " + tmpDir[li]);
            pairWriters.emplace_back([&, li]() {
                const Res& Rr = res[li];
                if(hlala_locus_write_pairs_file(acc[li].L, acc[li].li.n_clusters, Rr.order.data(), Rr.pNorm.data(), Rr.pairLL.data(), Rr.misAvg.data(), tmpDir[li].c_str()) != HLALA_OK)
                    { pairErr[li] = std::string("hlala_locus_write_pairs_file: ") + hlala_typer_last_error(); return; }
                write_one(li);
            });
            lociClock[5] += lap(tL);
        }
        timing.loci = lap(tLap);
        if(std::getenv("HLALA_HOST_DEBUG")) std::cerr << "host-debug: per-locus chain: filters " << lociClock[0] << ", buffers " << lociClock[1] << ", likelihoods + all pairs + call " << lociClock[2] << ", pair writer + k-mer lists " << lociClock[5] << " s\n";
        timing.kmers = 0;                                                          // (the k-mer questions are part of the per-locus chain since the reads are kept on the device)
        // ---- files: the writers of the loci finish; their directories are merged in locus order
        {
            auto tF = std::chrono::steady_clock::now();
            for(std::thread& x : pairWriters) x.join();
            for(const std::string& e : pairErr) if(!e.empty()) throw std::runtime_error(e);
            for(const std::string& e : ferr) if(!e.empty()) throw std::runtime_error(e);
            if(std::getenv("HLALA_HOST_DEBUG")) std::cerr << "host-debug: files: waiting for the writers of the loci " << lap(tF) << " s\n";
            for(size_t li = 0; li < acc.size(); li++) {
                DIR* dd = opendir(tmpDir[li].c_str());
                if(!dd) throw std::runtime_error("cannot open " + tmpDir[li]);
                std::vector<std::string> ents;
                while(dirent* e = readdir(dd)) { const std::string n = e->d_name; if(n != "." && n != "..") ents.push_back(n); }
                closedir(dd);
                for(const std::string& n : ents) {
                    const std::string from = tmpDir[li] + "/" + n, to = outputDirectory + "/" + n;
                    if(n == "R1_bestguess.txt" || n == "R1_bestguess_G.txt" || n == "histogram_matchesPerRead.txt") {
                        std::ifstream in(from.c_str(), std::ios::binary); std::ofstream app(to.c_str(), std::ios::binary | std::ios::app);
                        if(!in.is_open() || !app.is_open()) throw std::runtime_error("cannot append " + from + " to " + to);
                        app << in.rdbuf();
                        in.close(); ::unlink(from.c_str());
                    } else if(::rename(from.c_str(), to.c_str()) != 0) throw std::runtime_error("cannot move " + from + " to " + to);
                }
                ::rmdir(tmpDir[li].c_str());
                lociJoined += (lociJoined.empty() ? "" : ",") + acc[li].locus;
            }
        }
        tchk(hlala_typer_end_output(outputDirectory.c_str(), lociJoined.c_str(), 0), "hlala_typer_end_output");
        timing.files = lap(tLap);
        return out;
    }

    // graph level names (Graph::getOneLocusIDforLevel) and the gene list, for the caller's own files (reads_per_level.txt)
    int32_t n_levels() const { return hlala_typer_n_levels(t_); }
    std::string level_name(int32_t level) const { const char* n = hlala_typer_level_name(t_, level); return n ? n : ""; }
    bool has_locus(const std::string& locus) const { hlala_locus* L = nullptr; if(hlala_typer_locus(t_, locus.c_str(), 0, nullptr, &L) != HLALA_OK) return false; hlala_locus_free(L); return true; }

private:
    hlala_typer* t_ = nullptr; bool owns_ = true;
};

}  // namespace hla
}  // namespace host
}  // namespace hlala
