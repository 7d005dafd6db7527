// HLA-LA.cpp -- the `HLA-LA` host program of this repository: the process-level contract HLA-LA.pl relies on ("outer
// contract"), with the read-to-PRG alignment and the HLATyper scoring running on the GPU behind the C ABI of include/hlala_gpu.h.
//
//   HLA-LA --action testBinary                                   prints the line the installation check greps for   (HLA-LA.cpp:129-132)
//   HLA-LA --action prepareGraph --PRG_graph_dir G               leaves G/serializedGRAPH                          (HLA-LA.cpp:1341-1385; HLA-LA.pl:254-257)
//   HLA-LA --action HLA --maxThreads N --sampleID S --outputDirectory D --PRG_graph_dir G --FASTQU U --FASTQ1 R1 --FASTQ2 R2
//          --bwa_bin B --samtools_bin T --mapAgainstCompleteGenome 0|1 --longReads 0|ont2d|pacbio                  (HLA-LA.cpp:577-811; HLA-LA.pl:563)
//
// Arguments are `--name value` pairs, unknown names are ignored (HLA-LA.cpp:71-79).  Where the reference asserts or throws (abort /
// terminate), this program prints the message to stderr and exits with a non-zero status -- HLA-LA.pl treats any non-zero status as
// failure (:567-570).  Extra, optional arguments of this program: --devices <gpu,gpu,...> (or --device <gpu>): the batches of the sample are
// dealt round-robin to one context per listed GPU (a GPU may be listed twice: two contexts on it), results do not depend on the list;
// --decodeSlots <samples that decode at one time, default CPUs / 16>, --tailPool <k: GPU batches per launch of the widest DP classes>, --decodeThreads <host threads of the BAM decoder, default all>, --batchPairs <units per GPU batch>, --rngSeed <base of the end-cell draws>,
// --loci A,B,... (default: the reference's 17 loci, hla/HLATyper.cpp:42).  Several samples in one call (BASELINE config 4): comma-separated lists of
// equal length in --sampleID, --outputDirectory, --FASTQ1, --FASTQ2 (--FASTQU); sample i runs on device i % #devices, all samples side by side.
// Not rebuilt: the --BAM entry (the Perl driver never uses it: it extracts reads itself and passes FASTQ files), read simulation /
// validation actions, KIR.
#ifndef HLALA_HOST_TAIL_POOL_DEFAULT
#define HLALA_HOST_TAIL_POOL_DEFAULT 1
#endif
#include <sys/stat.h>
#include <sys/types.h>
#include <dirent.h>
#include <unistd.h>

#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <ctime>
#include <fstream>
#include <iostream>
#include <map>
#include <set>
#include <string>
#include <thread>
#include <vector>

#include "hlala_host.hpp"

using namespace hlala::host;

namespace {

bool fileExists(const std::string& p) { struct stat st; return stat(p.c_str(), &st) == 0 && S_ISREG(st.st_mode); }
bool directoryExists(const std::string& p) { struct stat st; return stat(p.c_str(), &st) == 0 && S_ISDIR(st.st_mode); }
void makeDir(const std::string& p) { if(mkdir(p.c_str(), 0775) != 0 && !directoryExists(p)) throw std::runtime_error("Cannot create directory " + p); }
// Utilities::make_or_clearDirectory: the directory exists and holds no regular files afterwards
void make_or_clearDirectory(const std::string& p)
{
    if(!directoryExists(p)) { makeDir(p); return; }
    DIR* d = opendir(p.c_str()); if(!d) throw std::runtime_error("Cannot open directory " + p);
    std::vector<std::string> files;
    while(dirent* e = readdir(d)) { std::string n = e->d_name; if(n != "." && n != ".." && fileExists(p + "/" + n)) files.push_back(p + "/" + n); }
    closedir(d);
    for(const std::string& f : files) if(unlink(f.c_str()) != 0) throw std::runtime_error("Cannot delete " + f);
}
std::string getFirstLine(const std::string& p) { std::ifstream f(p.c_str()); std::string l; std::getline(f, l); while(!l.empty() && (l.back() == '\n' || l.back() == '\r')) l.pop_back(); return l; }
std::string timestamp() { time_t t = time(nullptr); char b[64]; strftime(b, sizeof b, "[%Y-%m-%d %H:%M:%S] ", localtime(&t)); return b; }
bool StrtoB(const std::string& s) { return s == "1" || s == "true" || s == "TRUE" || s == "True"; }
void need(const std::map<std::string, std::string>& a, const char* k) { if(!a.count(k)) throw std::runtime_error(std::string("Missing argument --") + k); }
bool intervalsOverlap(int x1, int x2, int y1, int y2) { return x1 <= y2 && y1 <= x2; }          // Utilities.cpp:168-176 (closed intervals)

void run(const std::string& cmd)
{
    std::cerr << cmd << "\n" << std::flush;
    int rc = std::system(cmd.c_str());
    if(rc != 0) throw std::runtime_error("Command " + cmd + " returned code " + std::to_string(rc));
}

// mapper::bwa::BWAmapper (mapper/bwa/BWAmapper.cpp): the same command lines
struct BWAmapper {
    std::string bwa_bin, samtools_bin; int threads;
    bool ref_is_indexed(const std::string& ref) const { for(const char* s : {".sa", ".ann", ".bwt"}) if(!fileExists(ref + s)) return false; return true; }      // :53-65
    void make_sure_ref_is_indexed(const std::string& ref) const { if(!ref_is_indexed(ref)) run(bwa_bin + " index " + ref); }
    void sort_and_index(const std::string& outputBAM, const std::string& outputUnsorted) const
    {
        run(samtools_bin + " sort -@ " + std::to_string(threads) + " -o " + outputBAM + " " + outputUnsorted);
        if(!fileExists(outputBAM)) throw std::runtime_error("samtools sort did not produce " + outputBAM);
        run(samtools_bin + " index " + outputBAM);
        if(!fileExists(outputBAM + ".bai")) throw std::runtime_error("samtools index did not produce " + outputBAM + ".bai");
        unlink(outputUnsorted.c_str());
    }
    void prepare(const std::string& ref, const std::string& outputBAM, std::string& outputUnsorted) const
    {
        if(!fileExists(bwa_bin)) throw std::runtime_error("bwa binary not found: " + bwa_bin);
        if(!fileExists(samtools_bin)) throw std::runtime_error("samtools binary not found: " + samtools_bin);
        make_sure_ref_is_indexed(ref);
        if(fileExists(outputBAM)) unlink(outputBAM.c_str());
        outputUnsorted = outputBAM + ".unsorted";
        if(fileExists(outputUnsorted)) unlink(outputUnsorted.c_str());
        if(outputBAM.size() < 4 || outputBAM.substr(outputBAM.size() - 4) != ".bam") throw std::runtime_error("output BAM must end in .bam");
    }
    // BWAmapper::map, :178-243
    void map(const std::string& ref, const std::string& FASTQ1, const std::string& FASTQ2, const std::string& outputBAM, bool withA) const
    {
        if(!fileExists(FASTQ1) || !fileExists(FASTQ2)) throw std::runtime_error("FASTQ file not found: " + FASTQ1 + " / " + FASTQ2);
        std::string unsorted; prepare(ref, outputBAM, unsorted);
        run(bwa_bin + " mem -t" + std::to_string(threads) + " -M " + (withA ? "-a " : "") + ref + " " + FASTQ1 + " " + FASTQ2 + " | " + samtools_bin + " view -@ " +
            std::to_string(threads - 1) + " -Sb - > " + unsorted);
        sort_and_index(outputBAM, unsorted);
    }
    // BWAmapper::mapLong, :122-176
    void mapLong(const std::string& ref, const std::string& FASTQ, const std::string& outputBAM, bool withA, const std::string& longMode) const
    {
        if(!fileExists(FASTQ)) throw std::runtime_error("FASTQ file not found: " + FASTQ);
        std::string unsorted; prepare(ref, outputBAM, unsorted);
        run(bwa_bin + " mem -t" + std::to_string(threads) + " -x " + longMode + " -M " + (withA ? "-a " : "") + ref + " " + FASTQ + " | " + samtools_bin + " view -@ " +
            std::to_string(threads - 1) + " -Sb - > " + unsorted);
        sort_and_index(outputBAM, unsorted);
    }
};

int action_prepareGraph(const std::map<std::string, std::string>& arguments)
{
    need(arguments, "PRG_graph_dir");
    const std::string G = arguments.at("PRG_graph_dir");
    std::cout << "prepareGraph\n" << std::flush;
    std::cout << timestamp() << "Read graph from " << G << "\n" << std::flush;
    hlala_graph_file* g = nullptr;
    if(hlala_graph_load_text((G + "/PRG/graph.txt").c_str(), &g) != HLALA_OK) throw std::runtime_error(std::string("graph.txt: ") + hlala_loader_last_error());
    std::cout << timestamp() << "\tdone\n" << std::flush;
    hlala_graph_desc gd; hlala_graph_file_desc(g, &gd);
    // the reference writes two Boost archives, before and after computeGapEdgePaths; here the flattened arrays are the serialisation (the
    // gap-path index is rebuilt from them in well under a second at start-up), written under both names
    for(const char* name : {"/serializedGRAPH_preGapPathIndex", "/serializedGRAPH"}) {
        std::cout << timestamp() << "Now serialize graph to " << G << name << "\n" << std::flush;
        if(hlala_graph_cache_save(&gd, (G + name).c_str()) != HLALA_OK) { std::string e = hlala_loader_last_error(); hlala_graph_file_free(g); throw std::runtime_error("Cannot write " + G + name + ": " + e); }
        std::cout << timestamp() << "\tdone\n" << std::flush;
    }
    hlala_graph_file_free(g);
    return 0;
}

std::vector<std::string> split_list(const std::string& l)
{
    std::vector<std::string> out;
    for(size_t p = 0;;) { size_t q = l.find(',', p); out.push_back(l.substr(p, q == std::string::npos ? q : q - p)); if(q == std::string::npos) break; p = q + 1; }
    return out;
}

// what several samples of one call share: the graph directory as the aligner and as the typer read it (read once, read-only afterwards)
struct SharedGraph { std::shared_ptr<mapper::GraphDirectory> dir; std::unique_ptr<hla::HLATyper> typer;
                     // round 6: the device contexts of a call with several samples, one per listed device slot, created beside the first decodes and used by the samples in turn
                     std::vector<hlala_ctx*> ctx; std::vector<std::string> ctxErr; std::vector<std::thread> ctxThreads; std::mutex ctxMu; std::vector<char> ctxJoined;
                     hlala_ctx* context(size_t slot) { std::lock_guard<std::mutex> l(ctxMu); if(!ctxJoined[slot]) { if(ctxThreads[slot].joinable()) ctxThreads[slot].join(); ctxJoined[slot] = 1; } if(!ctxErr[slot].empty()) throw std::runtime_error(ctxErr[slot]); return ctx[slot]; }
                     ~SharedGraph() { for(std::thread& t : ctxThreads) if(t.joinable()) t.join(); for(hlala_ctx* c : ctx) if(c) hlala_destroy(c); } };

// Several samples in one call (round 6): who decodes and who holds a device when.  Within a sample the decode must be complete before the first batch is cut (a chain's
// random seed is its number in read-NAME order, mapper/processBAM.cpp:2024-2039); ACROSS samples nothing forbids decoding sample k + 1 on the host threads while the
// GPU aligns sample k.  Decodes start in sample order, as many at a time as the host has CPUs for (a decode keeps up to 32 threads busy); a device runs the alignment and
// typing of one sample at a time, in the order of the samples it was dealt.
struct SampleSchedule {
    std::mutex m; std::condition_variable cv;
    int decodeSlots = 1, decoding = 0; size_t nextDecode = 0;
    std::vector<size_t> nextOnDevice;
    void begin_decode(size_t sample) { std::unique_lock<std::mutex> l(m); cv.wait(l, [&] { return nextDecode == sample && decoding < decodeSlots; }); decoding++; nextDecode++; cv.notify_all(); }
    void end_decode() { std::lock_guard<std::mutex> l(m); decoding--; cv.notify_all(); }
    void begin_device(size_t slot, size_t rank) { std::unique_lock<std::mutex> l(m); cv.wait(l, [&] { return nextOnDevice[slot] == rank; }); }
    void end_device(size_t slot) { std::lock_guard<std::mutex> l(m); nextOnDevice[slot]++; cv.notify_all(); }
    // a sample that fails before its turn must not block the others
    void skip(size_t sample, size_t slot, size_t rank, bool decoded, bool onDevice) {
        std::unique_lock<std::mutex> l(m);
        if(!decoded) { cv.wait(l, [&] { return nextDecode >= sample; }); if(nextDecode == sample) nextDecode++; }
        if(!onDevice) { cv.wait(l, [&] { return nextOnDevice[slot] >= rank; }); if(nextOnDevice[slot] == rank) nextOnDevice[slot]++; }
        cv.notify_all();
    }
};
struct SampleTurn { SampleSchedule* sched; size_t sample, slot, rank; SampleTurn() : sched(nullptr), sample(0), slot(0), rank(0) {} SampleTurn(SampleSchedule* s, size_t a, size_t b, size_t c) : sched(s), sample(a), slot(b), rank(c) {} };

// CPUs this process may keep busy: the control group's quota when there is one (cgroup v2 cpu.max), else the hardware threads
static int cpu_budget()
{
    int n = (int)std::thread::hardware_concurrency(); if(n < 1) n = 1;
    std::ifstream f("/sys/fs/cgroup/cpu.max"); std::string q; long long per = 0;
    if(f.is_open() && (f >> q >> per) && q != "max" && per > 0) { const long long quota = std::atoll(q.c_str()); if(quota > 0) { const int c = (int)((quota + per - 1) / per); if(c >= 1 && c < n) n = c; } }
    return n;
}

int action_HLA_one(const std::map<std::string, std::string>& arguments, const std::vector<int>& devices, const SharedGraph* shared = nullptr, const SampleTurn* turn = nullptr)
{
    // (several samples per call: this sample's turns at the decoder and at its device; released on every way out)
    struct TurnGuard { const SampleTurn* t; bool decoding, decoded, onDevice, deviceDone;
        explicit TurnGuard(const SampleTurn* t_) : t(t_), decoding(false), decoded(false), onDevice(false), deviceDone(false) {}
        void begin_decode() { if(t) { t->sched->begin_decode(t->sample); decoding = true; } }
        void end_decode() { if(t && decoding) { t->sched->end_decode(); decoding = false; decoded = true; } }
        void begin_device() { if(t) { t->sched->begin_device(t->slot, t->rank); onDevice = true; } }
        void end_device() { if(t && onDevice && !deviceDone) { t->sched->end_device(t->slot); deviceDone = true; } }
        ~TurnGuard() { if(!t) return; if(decoding) { t->sched->end_decode(); decoded = true; } if(onDevice && !deviceDone) { t->sched->end_device(t->slot); deviceDone = true; } if(!decoded || !onDevice) t->sched->skip(t->sample, t->slot, t->rank, decoded, onDevice); } } turnGuard(turn);
    unsigned int maxThreads = 1;
    need(arguments, "sampleID"); need(arguments, "outputDirectory"); need(arguments, "PRG_graph_dir");
    if(!(arguments.count("BAM") || (arguments.count("FASTQ1") && arguments.count("FASTQ2")))) throw std::runtime_error("Please specify --BAM or --FASTQ1 / --FASTQ2");
    const std::string outputDirectory = arguments.at("outputDirectory"), PRG_graph_dir = arguments.at("PRG_graph_dir");
    if(arguments.count("FASTQ1")) { need(arguments, "mapAgainstCompleteGenome"); if(arguments.count("BAM")) throw std::runtime_error("--BAM and --FASTQ1 exclude each other"); }
    if(arguments.count("maxThreads")) { maxThreads = (unsigned)std::atoi(arguments.at("maxThreads").c_str()); if(maxThreads < 1) maxThreads = 1; std::cout << "Set maxThreads to " << maxThreads << "\n" << std::flush; }
    if(arguments.count("BAM"))
        throw std::runtime_error("--BAM is not supported by this build: HLA-LA.pl extracts the reads itself and calls --action HLA with --FASTQ1 / --FASTQ2 / --FASTQU");
    if(!directoryExists(outputDirectory)) makeDir(outputDirectory);

    const std::string BAM_remapped = outputDirectory + "/remapped_with_a.bam";
    const std::string PRGonlyReferenceGenomePath = PRG_graph_dir + "/mapping_PRGonly/referenceGenome.fa";
    std::string extendedReferenceGenomePath;
    if(fileExists(PRG_graph_dir + "/extendedReferenceGenomePath.txt")) extendedReferenceGenomePath = getFirstLine(PRG_graph_dir + "/extendedReferenceGenomePath.txt");
    else extendedReferenceGenomePath = PRG_graph_dir + "/extendedReferenceGenome/extendedReferenceGenome.fa";
    BWAmapper bwaMapper{arguments.at("bwa_bin"), arguments.at("samtools_bin"), (int)maxThreads};
    const bool remap_with_a = arguments.count("remap_with_a") ? StrtoB(arguments.at("remap_with_a")) : true;

    need(arguments, "FASTQ1"); need(arguments, "FASTQ2"); need(arguments, "longReads");
    std::string longReads = arguments.at("longReads");
    if(!(longReads == "0" || longReads == "ont2d" || longReads == "pacbio")) throw std::runtime_error("--longReads must be 0, ont2d or pacbio");
    if(longReads == "0") longReads = "";
    if(longReads.length()) need(arguments, "FASTQU");
    const bool mapAgainstCompleteGenome = StrtoB(arguments.at("mapAgainstCompleteGenome"));
    const std::string referenceGenomeForMapping = mapAgainstCompleteGenome ? extendedReferenceGenomePath : PRGonlyReferenceGenomePath;
    if(!fileExists(referenceGenomeForMapping)) throw std::runtime_error("Reference genome not found: " + referenceGenomeForMapping);
    if(longReads.length()) bwaMapper.mapLong(referenceGenomeForMapping, arguments.at("FASTQU"), BAM_remapped, remap_with_a, longReads);
    else bwaMapper.map(referenceGenomeForMapping, arguments.at("FASTQ1"), arguments.at("FASTQ2"), BAM_remapped, remap_with_a);
    std::cout << timestamp() << "Remapping done.\n" << std::flush;
    if(!fileExists(BAM_remapped) || !fileExists(BAM_remapped + ".bai")) throw std::runtime_error("Remapping did not produce " + BAM_remapped + " and its index");

    const int decodeThreads = arguments.count("decodeThreads") ? std::atoi(arguments.at("decodeThreads").c_str()) : 0;
    // --tailPool k: the broad / large / in-memory DP classes of k consecutive GPU batches of the sample run in one launch per class (include/hlala_gpu.h: hlala_set_tail_pool);
    // k + 1 batches are in flight per device.  Default HLALA_HOST_TAIL_POOL_DEFAULT; 1 = every batch runs its own tail (rounds 2-5)
    const int tailPool = arguments.count("tailPool") ? std::atoi(arguments.at("tailPool").c_str()) : HLALA_HOST_TAIL_POOL_DEFAULT;
    const auto tStart = std::chrono::steady_clock::now();
    const int32_t batchPairs = arguments.count("batchPairs") ? (int32_t)std::atol(arguments.at("batchPairs").c_str()) : (longReads.length() ? 65536 : 1048576);
    const uint32_t rngSeed = arguments.count("rngSeed") ? (uint32_t)std::strtoul(arguments.at("rngSeed").c_str(), nullptr, 10) : 0u;
    // long reads: columns of a read incl. the levels it skips (hlala_batch_create_unpaired)
    // the typer's view of the graph directory (level names of every segment file: millions of them) is read beside the graph, the contigs and the BAM
    std::unique_ptr<hla::HLATyper> typerPtr; std::string typerErr;
    std::thread typerThread([&]() {
        try {
            if(shared) typerPtr.reset(new hla::HLATyper(*shared->typer, hla::HLATyper::Borrow()));       // (several samples per call: read once by action_HLA)
            else typerPtr.reset(new hla::HLATyper(PRG_graph_dir, fileExists("hla_nom_g.txt") ? "hla_nom_g.txt" : ""));
        } catch(const std::exception& e) { typerErr = e.what(); } });
    struct Joiner { std::thread& t; ~Joiner() { if(t.joinable()) t.join(); } } typerJoin{typerThread};
    const std::shared_ptr<mapper::GraphDirectory> graphDirectory = shared ? shared->dir : std::make_shared<mapper::GraphDirectory>(PRG_graph_dir, mapAgainstCompleteGenome);
    mapper::processBAM BAMprocessor(graphDirectory, longReads.length() ? 16384 : 384, rngSeed, devices, decodeThreads);
    const double loadSeconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - tStart).count();
    turnGuard.begin_decode();
    std::cout << timestamp() << "Start seed extraction\n" << std::flush;
    const auto tOpen = std::chrono::steady_clock::now();
    const bool borrowed = turn && shared && !shared->ctx.empty();          // several samples: the call's contexts, taken when this sample's turn at the device comes
    BAMprocessor.openBAM(BAM_remapped, longReads.length() != 0, batchPairs, borrowed);
    if(!borrowed) BAMprocessor.set_tail_pool(tailPool);
    turnGuard.end_decode();
    const double openSeconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - tOpen).count();
    std::cout << timestamp() << "Seed extraction: " << BAMprocessor.n_units << " complete units, BAM decoded in " << BAMprocessor.decode_seconds << " s on " << BAMprocessor.decode_threads
              << " threads (index " << BAMprocessor.decode_phase_seconds[0] << ", inflate " << BAMprocessor.decode_phase_seconds[1] << ", parse " << BAMprocessor.decode_phase_seconds[2] << ", group "
              << BAMprocessor.decode_phase_seconds[3] << ", name sort " << BAMprocessor.decode_phase_seconds[4] << ", layout (sizes and offsets; the windows are filled batch by batch beside the GPU) " << BAMprocessor.decode_phase_seconds[5] << "); beside it: contexts on " << BAMprocessor.n_devices()
              << " device(s) ready after " << BAMprocessor.context_seconds << " s (of which " << BAMprocessor.directory_wait_seconds << " s waiting for the graph directory: reference intervals after " << graphDirectory->intervals_seconds
              << " s -- the decoder starts with them --, graph read after " << graphDirectory->graph_seconds << " s, translation tables after " << graphDirectory->contigs_seconds << " s, typer files beside them); seed extraction in all " << openSeconds << " s\n" << std::flush;
    if(!longReads.length() && !borrowed) std::cout << "Insert size: mean " << BAMprocessor.IS_mean << ", sd " << BAMprocessor.IS_sd << "\n" << std::flush;
    // the G-group table is looked up in the working directory, as the reference does (hla/HLATyper.cpp:4160-4166; HLA-LA.pl chdirs to the source directory)
    typerThread.join();
    if(!typerErr.empty()) throw std::runtime_error(typerErr);
    hla::HLATyper& HLAtyper = *typerPtr;
    std::vector<std::string> loci;
    if(arguments.count("loci")) loci = split_list(arguments.at("loci"));
    else for(const char* l : {"A", "B", "C", "DQA1", "DQB1", "DRB1", "DPA1", "DPB1", "DRA", "DRB3", "DRB4", "E", "F", "G", "H", "K", "V"}) {          // hla/HLATyper.cpp:42
        if(HLAtyper.has_locus(l)) loci.push_back(l); else std::cerr << "HLATypeInference(..): Locus " << l << ": no exon files in " << PRG_graph_dir << "/PRG -- skipped\n";
    }
    const std::string outputDirectory_for_HLA = outputDirectory + "/hla/";
    make_or_clearDirectory(outputDirectory + "/hla");                                                   // processBAM.cpp:1805-1806
    std::cout << timestamp() << "Alignment of " << BAMprocessor.n_units << (longReads.length() ? " reads" : " read pairs") << " in " << BAMprocessor.n_batches() << " GPU batch(es) on " << BAMprocessor.n_devices() << " device context(s)\n" << std::flush;
    double alignSeconds = 0; int64_t chainErrors = 0;
    turnGuard.begin_device();          // (the device may still be aligning the sample before this one; this sample's decode ran beside it)
    if(borrowed) { BAMprocessor.use_contexts(std::vector<hlala_ctx*>(1, const_cast<SharedGraph*>(shared)->context(turn->slot))); BAMprocessor.set_tail_pool(tailPool);
                   if(!longReads.length()) std::cout << "Insert size: mean " << BAMprocessor.IS_mean << ", sd " << BAMprocessor.IS_sd << "\n" << std::flush; }
    const auto tInfer = std::chrono::steady_clock::now();
    std::vector<hla::HLATyper::bestGuess> calls = HLAtyper.HLATypeInference(BAMprocessor, outputDirectory_for_HLA, loci, &alignSeconds, &chainErrors);
    const double inferSeconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - tInfer).count();
    const size_t pairs = longReads.length() ? 0 : (size_t)BAMprocessor.n_units, unpaired = longReads.length() ? (size_t)BAMprocessor.n_units : 0;
    std::cout << timestamp() << "Processed " << pairs << " protoSeeds (read pairs) / " << unpaired << " protoSeeds (unpaired long reads)\n" << std::flush;
    std::cout << "Speed: " << (alignSeconds > 0 ? (double)(pairs + unpaired) / alignSeconds : 0.0) << " protoSeeds (read pairs) per s" << "\n" << std::flush;      // :1894-1898
    // BAM bytes -> hla/*: seed extraction (decode, contexts, insert size) + alignment + typing (everything after the remapping)
    // Per SAMPLE: the decode, what openBAM does after it (page-locking the sample, insert-size estimate, hlala_set_insert_size), alignment and typing.
    // Per PROCESS (excluded): graph loading, and the part of context creation that outlasts the decode running beside it.
    { const double ctxBeyond = BAMprocessor.context_seconds > BAMprocessor.decode_seconds ? BAMprocessor.context_seconds - BAMprocessor.decode_seconds : 0.0;
      const double openSample = openSeconds - ctxBeyond;
      const double e2e = openSample + inferSeconds;
      std::cout << "End-to-end: " << (e2e > 0 ? (double)(pairs + unpaired) / e2e : 0.0) << " units per s (BAM decode " << BAMprocessor.decode_seconds << " s on " << BAMprocessor.decode_threads
                << " threads + page-locking and insert size " << openSample - BAMprocessor.decode_seconds << " s + alignment and typing " << inferSeconds << " s, of which the host spent " << BAMprocessor.layout_seconds() - BAMprocessor.decode_phase_seconds[5]
                << " s filling the windows of the batches beside the GPU; per process, not per sample: graph directory + context creation beyond the decode "
                << ctxBeyond << " s, reference intervals before it " << loadSeconds << " s; "
                << "whole action after the remapping: " << std::chrono::duration<double>(std::chrono::steady_clock::now() - tStart).count() << " s)\n" << std::flush; }
    std::cout << "Typing phases: batches (alignment, post-processing, exon positions) " << HLAtyper.timing.batches << " s, summary " << HLAtyper.timing.summary << " s, per-locus likelihoods and calls "
              << HLAtyper.timing.loci << " s (with the k-mer questions; the files of a locus are written beside the next locus), result files after the last locus " << HLAtyper.timing.files << " s\n" << std::flush;
    if(chainErrors) std::cerr << "WARNING: " << chainErrors << " alignments exceeded a device capacity; their read pairs are not used for typing\n";
    // reads_per_level.txt, processBAM.cpp:1902-1913
    {
        const std::vector<int32_t> cov = BAMprocessor.coverage();
        std::ofstream levels_stream((outputDirectory + "/reads_per_level.txt").c_str());
        if(!levels_stream.is_open()) throw std::runtime_error("Cannot open " + outputDirectory + "/reads_per_level.txt");
        for(size_t lI = 0; lI + 1 < (size_t)BAMprocessor.n_levels; lI++) levels_stream << lI << "\t" << HLAtyper.level_name((int32_t)lI) << "\t" << cov[lI] << "\n";
    }
    turnGuard.end_device();
    if(!fileExists(outputDirectory + "/hla/R1_bestguess.txt")) throw std::runtime_error("HLA type inference did not produce " + outputDirectory + "/hla/R1_bestguess.txt");
    for(const hla::HLATyper::bestGuess& g : calls) std::cout << "Locus " << g.locus << ": " << g.allele1 << " (Q1 " << g.Q1_allele1 << ") / " << g.allele2 << " (Q1 " << g.Q1_allele2 << ")\n";
    return 0;
}

// one sample on all listed devices, or several samples side by side, sample i on device i % #devices (BASELINE config 4: one sample per GPU; the
// result rows of every sample are its own files under its own output directory, as with one call per sample)
int action_HLA(const std::map<std::string, std::string>& arguments)
{
    std::vector<int> devices;
    if(arguments.count("devices")) for(const std::string& d : split_list(arguments.at("devices"))) devices.push_back(std::atoi(d.c_str()));
    else devices.push_back(arguments.count("device") ? std::atoi(arguments.at("device").c_str()) : 0);
    need(arguments, "sampleID");
    const std::vector<std::string> samples = split_list(arguments.at("sampleID"));
    if(samples.size() <= 1) return action_HLA_one(arguments, devices);
    std::vector<std::map<std::string, std::string>> per(samples.size(), arguments);
    for(const char* k : {"sampleID", "outputDirectory", "FASTQ1", "FASTQ2", "FASTQU"}) {
        if(!arguments.count(k)) continue;
        const std::vector<std::string> v = split_list(arguments.at(k));
        if(v.size() != samples.size()) throw std::runtime_error(std::string("--") + k + " must list one value per sample (" + std::to_string(samples.size()) + " samples)");
        for(size_t i = 0; i < samples.size(); i++) per[i][k] = v[i];
    }
    // the graph directory is read ONCE for all samples -- graph and contigs on this thread, the typer's view beside them -- and shared read-only
    // (eight samples on an eight-GPU node used to mean eight loads of the graph, eight of the 44 M translation lines and eight scans of the segment files)
    need(arguments, "PRG_graph_dir"); need(arguments, "mapAgainstCompleteGenome");
    const std::string PRG_graph_dir = arguments.at("PRG_graph_dir");
    SharedGraph shared; std::string typerErr;
    {
        const auto tLoad = std::chrono::steady_clock::now();
        ThreadJoiner tj;
        tj.start([&]() { try { shared.typer.reset(new hla::HLATyper(PRG_graph_dir, fileExists("hla_nom_g.txt") ? "hla_nom_g.txt" : "")); } catch(const std::exception& e) { typerErr = e.what(); } });
        shared.dir = std::make_shared<mapper::GraphDirectory>(PRG_graph_dir, StrtoB(arguments.at("mapAgainstCompleteGenome")));
        tj.join();
        if(!typerErr.empty()) throw std::runtime_error(typerErr);
        std::cout << timestamp() << "Graph directory read once for " << samples.size() << " samples in " << std::chrono::duration<double>(std::chrono::steady_clock::now() - tLoad).count() << " s\n" << std::flush;
    }
    std::vector<std::string> errs(samples.size());
    SampleSchedule sched;
    sched.nextOnDevice.assign(devices.size(), 0);
    { const int cpus = cpu_budget(); sched.decodeSlots = cpus / 16 > 0 ? cpus / 16 : 1; if(arguments.count("decodeSlots")) { const int v = std::atoi(arguments.at("decodeSlots").c_str()); if(v >= 1) sched.decodeSlots = v; } }
    // one context per device slot for the whole call, created on threads of their own while the first samples decode (hlala_host.hpp: processBAM::use_contexts)
    {
        const bool longR = arguments.count("longReads") && arguments.at("longReads") != "0";
        const uint32_t rngSeed = arguments.count("rngSeed") ? (uint32_t)std::strtoul(arguments.at("rngSeed").c_str(), nullptr, 10) : 0u;
        shared.ctx.assign(devices.size(), nullptr); shared.ctxErr.assign(devices.size(), ""); shared.ctxJoined.assign(devices.size(), 0);
        for(size_t d = 0; d < devices.size(); d++) shared.ctxThreads.emplace_back([&shared, &devices, d, longR, rngSeed]() {
            try {
                hlala_graph_desc gd; hlala_contigs_desc cd;
                hlala_graph_file_desc(shared.dir->graph(), &gd);
                if(hlala_contigs_file_desc(shared.dir->contigs(), &cd) != HLALA_OK) throw std::runtime_error(std::string("contigs: ") + hlala_loader_last_error());
                hlala_params pr{200.0, 35.0, rngSeed, longR ? 1 : 0, longR ? 16384 : 384, 0};
                if(hlala_create(&shared.ctx[d], devices[d], nullptr, &gd, &cd, &pr) != HLALA_OK) throw std::runtime_error(std::string("hlala_create: ") + hlala_last_error(nullptr));
            } catch(const std::exception& e) { shared.ctxErr[d] = e.what(); }
        });
    }
    std::vector<SampleTurn> turns(samples.size());
    for(size_t i = 0; i < samples.size(); i++) turns[i] = SampleTurn(&sched, i, i % devices.size(), i / devices.size());
    const auto tSamples = std::chrono::steady_clock::now();
    {
        ThreadJoiner th;
        for(size_t i = 0; i < samples.size(); i++) th.start([&, i]() {
            try { action_HLA_one(per[i], std::vector<int>(1, devices[i % devices.size()]), &shared, &turns[i]); } catch(const std::exception& e) { errs[i] = e.what(); }
        });
    }
    for(size_t i = 0; i < samples.size(); i++) if(!errs[i].empty()) throw std::runtime_error("sample " + samples[i] + ": " + errs[i]);
    std::cout << timestamp() << "Processed " << samples.size() << " samples on " << devices.size() << " device(s)\n" << std::flush;
    std::cout << "Samples: " << samples.size() << " on " << devices.size() << " device(s), " << sched.decodeSlots << " decoding at a time, in " << std::chrono::duration<double>(std::chrono::steady_clock::now() - tSamples).count()
              << " s after the graph directory (remapping, decode of sample k + 1 beside the alignment of sample k, contexts, alignment, typing, result files)\n" << std::flush;
    return 0;
}

}  // namespace

int main(int argc, char* argv[])
{
    try {
        // the BAM decoder's worker threads at lower priority (the library reads the variable once; an explicit setting wins): the threads that create contexts, feed the GPU and
        // write result files stay ahead of the decoder's, which outnumber the CPUs a control group grants -- four samples in one call 1.98 -> 2.10 M pairs/s, one sample 1.99 -> 2.07 M
        // on a 16-CPU box (profiles/r06_experiments.txt 16)
        (void)setenv("HLALA_BAM_NICE", "10", 0);
        std::vector<std::string> ARG(argv + 1, argv + argc);
        std::map<std::string, std::string> arguments;
        for(unsigned int i = 0; i < ARG.size(); i++)
            if((ARG.at(i).length() > 2) && (ARG.at(i).substr(0, 2) == "--")) {                          // HLA-LA.cpp:71-79 (a trailing name without a value throws there too)
                if(i + 1 >= ARG.size()) throw std::runtime_error("Argument " + ARG.at(i) + " has no value");
                arguments[ARG.at(i).substr(2)] = ARG.at(i + 1);
            }
        // the start-up self test of the reference, HLA-LA.cpp:94-102
        if(intervalsOverlap(1, 10, 11, 20) || intervalsOverlap(5, 11, 1, 4) || !intervalsOverlap(5, 11, 8, 11) || !intervalsOverlap(8, 11, 1, 9) || !intervalsOverlap(8, 11, 9, 10) ||
           !intervalsOverlap(9, 10, 8, 11) || !intervalsOverlap(1, 10, 2, 3) || !intervalsOverlap(2, 3, 1, 10)) throw std::runtime_error("intervalsOverlap self test failed");
        if(arguments.count("action") == 0) {
            std::cerr << "\n\nMissing --action parameter. Please don't try calling me directly; use HLA-LA.pl instead (see documentation on GitHub).\n" << std::endl;
            throw std::runtime_error("Missing arguments -- see above.");
        }
        const std::string action = arguments.at("action");
        const std::set<std::string> noBinariesRequired = {"prepareGraph", "testBinary"};
        if(noBinariesRequired.count(action) == 0 && !arguments.count("bwa_bin")) throw std::runtime_error("Please specify arguments --bwa_bin");
        if(noBinariesRequired.count(action) == 0 && !arguments.count("samtools_bin")) throw std::runtime_error("Please specify arguments --samtools_bin");
        if(!std::system(NULL)) { std::cerr << "\n\nMissing shell - std::system(NULL) has returned a 0 value.\n" << std::endl; throw std::runtime_error("Missing shell"); }
        if(action == "testBinary") { std::cout << "\nHLA*LA binary functional!\n\n"; return 0; }
        if(action == "prepareGraph") return action_prepareGraph(arguments);
        if(action == "HLA") return action_HLA(arguments);
        throw std::runtime_error("Action " + action + " is not part of this build (available: HLA, prepareGraph, testBinary)");
    } catch(const std::exception& e) {
        std::cerr << "HLA-LA: " << e.what() << "\n" << std::flush;
        return 1;
    }
}
