// graphm.cpp -- "Graph M": MHC-scale stand-in for PRG_MHC_GRCh38_withIMGT plus a read-pair /
// bwa-like seed simulator on it.  TEST / BENCH INFRASTRUCTURE, not part of the product; nothing here reads /root/reference.
//
// World: a chain of SEGMENTS that meet in single nodes (as the segment files of a PRG do):
//   backbone segments  - `n_backbone` haplotypes at pairwise divergence `backbone_div` (70 % substitutions, 30 % single-level
//                        gaps as in simpleGraphSimulator.cpp:146-231) with multi-level gap stretches covering `gap_stretch_frac`
//                        of the levels;
//   gene windows       - `win_len_min..win_len_max` levels carrying `alleles_min..alleles_max` allele paths: mosaics of a few dozen
//                        founder lineages (polymorphic sites at `exon_site_density` inside exons, `intron_site_density` outside,
//                        founder-specific deletions) with private substitutions on top; rows 0..n_backbone-1 of a window are the
//                        alleles the backbone haplotypes carry through it.
// Graph of a segment: the rule Graph::buildFromHaplotypes(hp, ..., want_suffix_length = 10) applies when it joins edge groups
// (Graph/Graph.cpp:567-1140; called with 10 at Graph/graphSimulator/simpleGraphSimulator.cpp:268): haplotypes that leave one node
// with the same symbol share an edge; two edges end in the same node iff the SETS of upcoming symbol strings of their haplotypes
// (read on until 10 non-gap symbols were seen, at most 100 symbols) are equal and none of them begins with a gap.  Deviations,
// deliberate: segments end in one node (the reference's END_PUFFER tail leaves the last 10 levels unjoined), the string of a
// haplotype ends at ITS tenth non-gap symbol (the reference extends all strings of a comparison to the longest need).
// Node / edge creation order (= the canonical order of the C ABI): level-major, then first-haplotype order.
//
// Reads: pairs from a backbone haplotype or, with probability `frac_gene`, from ANY allele row of a gene window (most rows have
// no contig of their own, so their reads carry variants the contig they were "mapped" to lacks).  Per-position quality and
// correctness probabilities come from an empirical matrix (the reference's simulator/predefinedQualityMatrices/I101_NA12878.txt,
// kept as data under tools/data/) stretched to the read length by the nearest-index rule of readSimulator.cpp:193-205; insertions
// and deletions ~ Poisson(1e-4) per base each (:218-221, :449-490).  Seeds: the read placed on candidate contigs through the shared level
// coordinate (CIGAR with M/I/D from the column correspondence), soft-clipped Beta(1,4)*clip_max bases per end; primary = best
// score; AS-descending order.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <map>
#include <string>
#include <unordered_map>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef uint8_t u8;
typedef uint64_t u64;

namespace {

struct Rng {
    u64 s[4];
    static u64 splitmix(u64& x) { u64 z = (x += 0x9E3779B97F4A7C15ull); z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }
    explicit Rng(u64 seed) { u64 x = seed; for(int i = 0; i < 4; i++) s[i] = splitmix(x); }
    static u64 rotl(u64 x, int k) { return (x << k) | (x >> (64 - k)); }
    u64 next() { u64 r = rotl(s[1] * 5, 7) * 9, t = s[1] << 17; s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3]; s[2] ^= t; s[3] = rotl(s[3], 45); return r; }
    double uni() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); }
    int below(int n) { return n <= 1 ? 0 : (int)(next() % (u64)n); }
    long long below64(long long n) { return n <= 1 ? 0 : (long long)(next() % (u64)n); }
    int poisson(double lam) { if(lam <= 0) return 0; return poissonL(exp(-lam)); }
    int poissonL(double L) { double p = 1.0; int k = 0; do { k++; p *= uni(); } while(p > L); return k - 1; }      // L = exp(-lambda)
    double normal() { double u1 = uni(), u2 = uni(); if(u1 < 1e-300) u1 = 1e-300; return sqrt(-2.0 * log(u1)) * cos(6.283185307179586 * u2); }
    int geometric(double p) { int k = 1; while(uni() >= p && k < 1000) k++; return k; }
    // Beta(1, 4): 1 - U^(1/4)
    double beta14() { return 1.0 - pow(uni(), 0.25); }
};

const u8 NUC[4] = {'A', 'C', 'G', 'T'};
inline u8 other_base(Rng& r, u8 b) { for(;;) { u8 c = NUC[r.below(4)]; if(c != b) return c; } }

struct Segment {
    int nh = 0, len = 0;            // haplotype rows, columns (= levels spanned; the segment has len + 1 node levels, the last shared)
    std::vector<u8> M;              // [nh * len]
    bool gene = false;
    std::vector<u8> exon;           // [len] gene windows: 1 = exon column
    // graph of the segment
    std::vector<int> nodes_per_level;                // [len + 1]
    std::vector<int> e_level, e_from, e_to;          // ranks within their levels
    std::vector<u8> e_label;
    long long level0 = 0;           // global level of column 0
};

// ---------------------------------------------------------------------------------------- segment graph (suffix rule)
void build_segment_graph(Segment& S, int want)
{
    const int nh = S.nh, len = S.len;
    const u8* M = S.M.data();
    S.nodes_per_level.assign(len + 1, 0);
    S.nodes_per_level[0] = 1;
    std::vector<int> node_of(nh, 0), grp(nh), grp_node;
    // rolling hash of the upcoming-symbol string of every haplotype: symbols [a, e) with `want` non-gap symbols (or the end)
    const u64 B = 0x100000001B3ull | 1ull;
    u64 Binv = 1; { u64 x = B; for(int i = 0; i < 6; i++) { Binv = x * (2 - B * x); x = Binv; } }   // Newton: inverse of B mod 2^64
    // (start: x = B is an inverse mod 8 for odd B; each step doubles the bits)
    std::vector<u64> H(nh, 0), PW(nh, 1); std::vector<int> E(nh, 0), NG(nh, 0);
    auto extend = [&](int h, int a) {
        const u8* row = M + (size_t)h * len;
        while(NG[h] < want && E[h] < len && E[h] - a < 100) { u8 c = row[E[h]]; H[h] += (u64)(c + 1) * PW[h]; PW[h] *= B; if(c != '_') NG[h]++; E[h]++; }
    };
    for(int h = 0; h < nh; h++) { E[h] = 1; extend(h, 1); }    // strings that start at column 1 (the first target level)
    // note: E starts at 1 with an empty string: H = 0, PW = 1
    std::vector<std::pair<u64, u64>> pairs; pairs.reserve(nh);
    std::vector<int> key2grp;
    for(int l = 0; l < len; l++) {
        const int nPrev = S.nodes_per_level[l];
        // ---- edge groups: (node, symbol) in first-haplotype order
        key2grp.assign((size_t)nPrev * 8, -1);
        int nG = 0;
        std::vector<int> g_from; std::vector<u8> g_sym;
        for(int h = 0; h < nh; h++) {
            u8 c = M[(size_t)h * len + l];
            int code = c == 'A' ? 0 : c == 'C' ? 1 : c == 'G' ? 2 : c == 'T' ? 3 : c == '_' ? 4 : 5;
            int& g = key2grp[(size_t)node_of[h] * 8 + code];
            if(g < 0) { g = nG++; g_from.push_back(node_of[h]); g_sym.push_back(c); }
            grp[h] = g;
        }
        grp_node.assign(nG, -1);
        int nNext = 0;
        if(l == len - 1) { for(int g = 0; g < nG; g++) grp_node[g] = 0; nNext = 1; }
        else {
            // signature of a group = its sorted set of upcoming strings; a string that begins with a gap, or that cannot
            // collect `want` non-gap symbols within 100, makes the group unjoinable
            pairs.clear();
            std::vector<u8> unjoin(nG, 0);
            for(int h = 0; h < nh; h++) {
                const u8* row = M + (size_t)h * len;
                if(row[l + 1] == '_') unjoin[grp[h]] = 1;
                if(NG[h] < want && E[h] < len) unjoin[grp[h]] = 1;
                pairs.emplace_back((u64)grp[h], H[h] ^ ((u64)(E[h] - (l + 1)) << 56));
            }
            std::sort(pairs.begin(), pairs.end());
            pairs.erase(std::unique(pairs.begin(), pairs.end()), pairs.end());
            std::vector<u64> sig(nG, 0x9E3779B97F4A7C15ull);
            for(auto& pr : pairs) { u64& s = sig[pr.first]; s = (s ^ pr.second) * 0xFF51AFD7ED558CCDull; s ^= s >> 29; }
            std::unordered_map<u64, int> sig2node;
            for(int g = 0; g < nG; g++) {
                if(unjoin[g]) { grp_node[g] = nNext++; continue; }
                auto it = sig2node.find(sig[g]);
                if(it == sig2node.end()) { sig2node.emplace(sig[g], nNext); grp_node[g] = nNext++; } else grp_node[g] = it->second;
            }
        }
        S.nodes_per_level[l + 1] = nNext;
        for(int g = 0; g < nG; g++) { S.e_level.push_back(l); S.e_from.push_back(g_from[g]); S.e_to.push_back(grp_node[g]); S.e_label.push_back(g_sym[g]); }
        for(int h = 0; h < nh; h++) node_of[h] = grp_node[grp[h]];
        // ---- advance the strings to start at column l + 2
        if(l + 2 <= len) for(int h = 0; h < nh; h++) {
            const int a = l + 1;
            if(E[h] > a) { u8 c = M[(size_t)h * len + a]; H[h] = (H[h] - (u64)(c + 1)) * Binv; PW[h] *= Binv; if(c != '_') NG[h]--; }
            else E[h] = a + 1;        // empty string: keep it empty at the new start
            extend(h, a + 1);
        }
    }
}

struct Params {
    u64 seed; long long n_levels; int n_backbone; double backbone_div; double gap_stretch_frac;
    int n_windows, win_len_min, win_len_max, alleles_min, alleles_max;
    double exon_site_density, intron_site_density, private_rate; int suffix_len; int contigs_per_window; int threads; double hyper_site_density;
};

void make_backbone_segment(Segment& S, Rng& r, const Params& P, int len)
{
    const int nh = P.n_backbone;
    S.nh = nh; S.len = len; S.gene = false; S.M.resize((size_t)nh * len);
    std::vector<u8> sc(len);
    for(int i = 0; i < len; i++) sc[i] = NUC[r.below(4)];
    for(int h = 0; h < nh; h++) {
        u8* row = S.M.data() + (size_t)h * len;
        memcpy(row, sc.data(), len);
        if(h == 0) continue;
        // geometric skipping over the events of this haplotype (rate backbone_div / 2 per column)
        const double rate = P.backbone_div * 0.5;
        if(rate > 0) for(long long i = 12; i < len - 12;) {
            double u = r.uni(); long long skip = (long long)floor(log(u < 1e-300 ? 1e-300 : u) / log(1.0 - rate));
            i += skip; if(i >= len - 12) break;
            if(r.uni() < 0.3) row[i] = '_'; else row[i] = other_base(r, sc[i]);
            i++;
        }
    }
    // gap stretches: 3 + Poisson(10) levels, a random non-empty proper subset of the haplotypes skips them
    if(P.gap_stretch_frac > 0 && nh > 1) {
        const double startRate = P.gap_stretch_frac / 13.0;
        for(long long i = 16; i < len - 40;) {
            double u = r.uni(); long long skip = (long long)floor(log(u < 1e-300 ? 1e-300 : u) / log(1.0 - startRate));
            i += skip; if(i >= len - 40) break;
            int L = 3 + r.poisson(10.0); if(i + L >= len - 14) break;
            u64 mask = 0; while(mask == 0 || mask == ((1ull << nh) - 1)) mask = r.next() & ((1ull << nh) - 1);
            for(int h = 0; h < nh; h++) if(mask >> h & 1) memset(S.M.data() + (size_t)h * len + i, '_', L);
            i += L + 12;
        }
    }
}

void make_gene_segment(Segment& S, Rng& r, const Params& P, int len, int nAll)
{
    S.nh = nAll; S.len = len; S.gene = true; S.M.resize((size_t)nAll * len); S.exon.assign(len, 0);
    std::vector<u8> cons(len);
    for(int i = 0; i < len; i++) cons[i] = NUC[r.below(4)];
    // exons: ~270 columns each, introns 200-900 columns; the first exon is HYPERVARIABLE (mask value 2: most columns polymorphic with
    // common variants, like exon 2 of a classical HLA gene), the others carry mostly rare variants
    { int ne = 0; for(int p = 100 + r.below(300); p + 300 < len - 20;) { int el = 200 + r.below(140); for(int i = p; i < p + el && i < len - 14; i++) S.exon[i] = ne == 0 ? 2 : 1; ne++; p += el + 200 + r.below(700); } }
    int F = nAll / 8; if(F < 16) F = 16; if(F > 600) F = 600; if(F > nAll) F = nAll;     // founder lineages (allele families)
    std::vector<u8> FM((size_t)F * len);
    for(int f = 0; f < F; f++) memcpy(FM.data() + (size_t)f * len, cons.data(), len);
    for(int i = 12; i < len - 12; i++) {
        double dens = S.exon[i] == 2 ? P.hyper_site_density : S.exon[i] ? P.exon_site_density : P.intron_site_density;
        double u = r.uni();
        if(u < dens) {
            u8 alt = other_base(r, cons[i]); double u3 = r.uni(); double f = S.exon[i] == 2 ? 0.05 + 0.4 * u3 : 0.01 + 0.15 * u3 * u3 * u3; bool any = false;      // mostly rare variants
            for(int k = 0; k < F; k++) if(r.uni() < f) { FM[(size_t)k * len + i] = alt; any = true; }
            if(!any) FM[(size_t)r.below(F) * len + i] = alt;
            if(r.uni() < 0.15) { u8 alt2 = other_base(r, cons[i]); FM[(size_t)r.below(F) * len + i] = alt2; }    // a third variant now and then
        } else if(u < dens + 0.002) {
            int L = S.exon[i] ? 1 + r.below(3) : (r.uni() < 0.8 ? 1 + r.below(4) : 5 + r.below(26));
            if(i + L >= len - 14) continue;
            double f = 0.05 + 0.3 * r.uni(); bool any = false;
            for(int k = 0; k < F; k++) if(r.uni() < f) { memset(FM.data() + (size_t)k * len + i, '_', L); any = true; }
            if(!any) memset(FM.data() + (size_t)r.below(F) * len + i, '_', L);
            i += L + 2;
        }
    }
    for(int a = 0; a < nAll; a++) {
        u8* row = S.M.data() + (size_t)a * len;
        int f = r.below(F);
        for(int i = 0; i < len;) {
            // stay on founder f for a geometric stretch (mean 400), never switching inside a deletion
            int run = 1 + (int)floor(log(std::max(1e-300, r.uni())) / log(1.0 - 1.0 / 400.0));
            int e = std::min(len, i + run);
            while(e < len && FM[(size_t)f * len + e] == '_') e++;
            memcpy(row + i, FM.data() + (size_t)f * len + i, e - i);
            i = e; f = r.below(F);
        }
        // private substitutions
        const double pr = P.private_rate;
        if(pr > 0) for(long long i = 12; i < len - 12;) {
            double u = r.uni(); long long skip = (long long)floor(log(u < 1e-300 ? 1e-300 : u) / log(1.0 - pr));
            i += skip; if(i >= len - 12) break;
            if(row[i] != '_' && (S.exon[i] || r.uni() < 0.5)) row[i] = other_base(r, row[i]);
            i++;
        }
    }
}

struct Contig { std::vector<u8> seq; std::vector<int> level; int window = -1; int row = -1; };

struct World {
    Params P;
    std::vector<Segment> segs;
    long long L = 0, N = 0, E = 0;
    std::vector<int> node_level, edge_from, edge_to; std::vector<u8> edge_label;
    std::vector<long long> level_off;
    std::vector<Contig> contigs;
    std::vector<int> win_seg;          // segment index of window w
    int max_nodes_per_level = 0;
    std::string err;
};

struct QualMatrix {
    int len = 0;                                       // native read length of the matrix
    std::vector<std::vector<std::pair<u8, double>>> freq;   // per position: (quality char, cumulative frequency)
    std::vector<std::vector<double>> correct;          // per position: EmpiricalCorrect by quality char [256]
    std::vector<double> indel;                         // per position
};

struct BatchParams {
    u64 seed; int n_pairs, read_len; double jump_mean, jump_sd; int clip_max; double p_no_clip;
    double frac_gene; double p_secondary; int max_secondary; double p_random_secondary; double p_wrong_strand; double p_flip; int gene_candidates;
};

struct Batch {
    int n_pairs = 0, read_len = 0;
    std::vector<int> read_off, chain_off, read_primary, chain_contig, chain_pos, chain_as, cigar_off, truth_level, read_window;
    std::vector<u8> read_bases, read_quals, chain_reverse;
    std::vector<uint32_t> cigar;
};

bool load_matrix(const char* path, QualMatrix& Q, std::string& err)
{
    FILE* f = fopen(path, "r");
    if(!f) { err = std::string("cannot open ") + path; return false; }
    char line[512]; int ln = 0;
    std::map<int, std::map<int, std::vector<std::pair<u8, double>>>> fr;     // len -> pos -> (q, N)
    std::map<int, std::map<int, std::map<int, double>>> co;
    while(fgets(line, sizeof line, f)) {
        if(ln++ == 0) continue;
        int rl, pos; char q; double N, ex, em;
        // fields are tab-separated; the quality is one raw character
        char* p = line; rl = (int)strtol(p, &p, 10); if(*p != '\t') continue; p++;
        q = *p; p++; if(*p != '\t') continue; p++;
        pos = (int)strtol(p, &p, 10); N = strtod(p, &p); ex = strtod(p, &p); em = strtod(p, &p); (void)ex;
        fr[rl][pos].emplace_back((u8)q, N); co[rl][pos][(u8)q] = em;
    }
    fclose(f);
    if(fr.empty()) { err = "empty quality matrix"; return false; }
    int rl = fr.begin()->first; Q.len = rl;
    Q.freq.resize(rl); Q.correct.assign(rl, std::vector<double>(256, 1.0)); Q.indel.assign(rl, 1e-4);
    for(int p = 0; p < rl; p++) {
        auto& v = fr[rl][p]; double tot = 0; for(auto& x : v) tot += x.second;
        if(tot <= 0) { err = "quality matrix position without data"; return false; }
        double cum = 0;
        for(auto& x : v) if(x.second > 0) { cum += x.second / tot; Q.freq[p].emplace_back(x.first, cum); Q.correct[p][x.first] = co[rl][p][x.first]; }
        Q.freq[p].back().second = 1.0;
    }
    return true;
}

// ------------------------------------------------------------------------------- alignment of a simulated read to a contig
struct Aln { int contig, pos, as; bool rev; bool primary; std::vector<std::pair<int, char>> ops; };

// read bases rb[0..n) with truth levels tl (-1 = inserted base); contig with sorted levels.  Returns false if nothing aligns.
bool align_by_levels(const Contig& C, const u8* rb, const int* tl, int n, int clipL, int clipR, Aln& out)
{
    std::vector<char> op; op.reserve(n + 16);            // one char per operation unit
    int hint = -1;
    int cur = -1, startIdx = -1;                         // cur: next contig index to account for (-1: no contig position consumed yet)
    std::vector<int> mIdx; mIdx.reserve(n + 16);         // contig index for M / D units, -1 for I
    const int* lv = C.level.data(); const int cl = (int)C.level.size();
    for(int i = 0; i < n; i++) {
        if(tl[i] < 0) { op.push_back('I'); mIdx.push_back(-1); continue; }
        // levels of a read ascend: search from the previous hit (a few steps), binary search only for the first base
        int idx;
        if(hint < 0) idx = (int)(std::lower_bound(lv, lv + cl, tl[i]) - lv);
        else { idx = hint; int steps = 0; while(idx < cl && lv[idx] < tl[i] && steps < 64) { idx++; steps++; } if(idx < cl && lv[idx] < tl[i]) idx = (int)(std::lower_bound(lv + idx, lv + cl, tl[i]) - lv); }
        hint = idx;
        bool has = idx < cl && lv[idx] == tl[i];
        if(cur >= 0) for(int k = cur; k < idx; k++) { op.push_back('D'); mIdx.push_back(k); }      // contig bases the read skips
        if(has) { op.push_back('M'); mIdx.push_back(idx); cur = idx + 1; if(startIdx < 0) startIdx = idx; }
        else { op.push_back('I'); mIdx.push_back(-1); if(cur >= 0) cur = idx; }                     // the contig has a gap at this level
    }
    if(startIdx < 0) return false;
    // trim: leading / trailing non-M units become clipped read bases; then apply the soft clips
    int nu = (int)op.size();
    int a = 0; while(a < nu && op[a] != 'M') a++;
    int b = nu; while(b > a && op[b - 1] != 'M') b--;
    int readL = 0; for(int k = 0; k < a; k++) if(op[k] == 'I') readL++;
    int readR = 0; for(int k = b; k < nu; k++) if(op[k] == 'I') readR++;
    // extend the clips to clipL / clipR read bases
    while(readL < clipL && a < b) { if(op[a] != 'D') readL++; a++; }
    while(a < b && op[a] != 'M') { if(op[a] == 'I') readL++; a++; }
    while(readR < clipR && b > a) { if(op[b - 1] != 'D') readR++; b--; }
    while(b > a && op[b - 1] != 'M') { if(op[b - 1] == 'I') readR++; b--; }
    if(b - a < 20) return false;
    out.pos = mIdx[a];
    out.ops.clear();
    if(readL) out.ops.emplace_back(readL, 'S');
    int score = 0, ri = readL;
    for(int k = a; k < b;) {
        int e = k; while(e < b && op[e] == op[k]) e++;
        out.ops.emplace_back(e - k, op[k]);
        if(op[k] == 'M') { for(int q = k; q < e; q++) { score += (C.seq[mIdx[q]] == rb[ri]) ? 1 : -4; ri++; } }
        else { score -= 6 + (e - k); if(op[k] == 'I') ri += e - k; }
        k = e;
    }
    if(readR) out.ops.emplace_back(readR, 'S');
    out.as = score;
    return true;
}

}  // namespace

extern "C" {

struct gm_params { u64 seed; long long n_levels; int n_backbone; double backbone_div; double gap_stretch_frac;
                   int n_windows, win_len_min, win_len_max, alleles_min, alleles_max;
                   double exon_site_density, intron_site_density, private_rate; int suffix_len; int contigs_per_window; int threads; double hyper_site_density; };
struct gm_batch_params { u64 seed; int n_pairs, read_len; double jump_mean, jump_sd; int clip_max; double p_no_clip;
                         double frac_gene; double p_secondary; int max_secondary; double p_random_secondary; double p_wrong_strand; double p_flip; int gene_candidates; };

static std::string g_err;
const char* gm_last_error() { return g_err.c_str(); }

void* gm_world_create(const gm_params* pp)
{
    World* W = new World();
    Params& P = W->P;
    P.seed = pp->seed; P.n_levels = pp->n_levels; P.n_backbone = pp->n_backbone; P.backbone_div = pp->backbone_div; P.gap_stretch_frac = pp->gap_stretch_frac;
    P.n_windows = pp->n_windows; P.win_len_min = pp->win_len_min; P.win_len_max = pp->win_len_max; P.alleles_min = pp->alleles_min; P.alleles_max = pp->alleles_max;
    P.exon_site_density = pp->exon_site_density; P.intron_site_density = pp->intron_site_density; P.private_rate = pp->private_rate;
    P.suffix_len = pp->suffix_len; P.contigs_per_window = pp->contigs_per_window; P.threads = pp->threads; P.hyper_site_density = pp->hyper_site_density;
    if(P.n_backbone < 1 || P.n_backbone > 60 || P.n_windows < 0 || P.win_len_min < 200) { g_err = "bad parameters"; delete W; return nullptr; }
    Rng top(P.seed);
    // ---- layout: backbone, window, backbone, ..., window, backbone
    std::vector<int> wlen(P.n_windows), wall(P.n_windows);
    long long wsum = 0;
    for(int w = 0; w < P.n_windows; w++) {
        wlen[w] = P.win_len_min + top.below(P.win_len_max - P.win_len_min + 1);
        double la = log((double)P.alleles_min), lb = log((double)P.alleles_max);
        wall[w] = (int)floor(exp(la + (lb - la) * top.uni()) + 0.5);
        if(wall[w] < P.n_backbone) wall[w] = P.n_backbone;
        wsum += wlen[w];
    }
    long long rest = P.n_levels - wsum;
    int nb = P.n_windows + 1;
    if(rest < (long long)nb * 64) { g_err = "n_levels too small for the windows"; delete W; return nullptr; }
    int nSeg = 2 * P.n_windows + 1;
    W->segs.resize(nSeg);
    std::vector<int> slen(nSeg);
    for(int s = 0; s < nSeg; s++) {
        if(s & 1) slen[s] = wlen[s / 2];
        else { long long base = rest / nb; slen[s] = (int)(base + ((s / 2) < (rest % nb) ? 1 : 0)); }
    }
    std::vector<u64> sseed(nSeg); for(int s = 0; s < nSeg; s++) sseed[s] = top.next();
    // very long backbone stretches are cut into pieces of <= 2^18 columns so that the work spreads over the threads
    // (pieces meet in single nodes like any other segments)
    {
        std::vector<Segment> segs2; std::vector<int> len2; std::vector<u64> seed2; std::vector<int> isWin;
        for(int s = 0; s < nSeg; s++) {
            if(s & 1) { len2.push_back(slen[s]); seed2.push_back(sseed[s]); isWin.push_back(s / 2); continue; }
            int left = slen[s], piece = 0;
            while(left > 0) { int l = left > (1 << 18) + 4096 ? (1 << 18) : left; len2.push_back(l); seed2.push_back(sseed[s] + 0x1234567ull * (u64)(piece++)); isWin.push_back(-1); left -= l; }
        }
        nSeg = (int)len2.size(); W->segs.clear(); W->segs.resize(nSeg); slen = len2; sseed = seed2;
        W->win_seg.assign(P.n_windows, -1);
        for(int s = 0; s < nSeg; s++) if(isWin[s] >= 0) W->win_seg[isWin[s]] = s;
#ifdef _OPENMP
        if(P.threads > 0) omp_set_num_threads(P.threads);
#endif
#pragma omp parallel for schedule(dynamic, 1)
        for(int s = 0; s < nSeg; s++) {
            Rng r(sseed[s]);
            if(isWin[s] >= 0) make_gene_segment(W->segs[s], r, P, slen[s], wall[isWin[s]]);
            else make_backbone_segment(W->segs[s], r, P, slen[s]);
            build_segment_graph(W->segs[s], P.suffix_len);
        }
    }
    // ---- global numbering
    long long L = 1; for(auto& S : W->segs) { S.level0 = L - 1; L += S.len; }
    W->L = L;
    W->level_off.assign(L + 1, 0);
    for(auto& S : W->segs) for(int l = 0; l <= S.len; l++) { long long gl = S.level0 + l; int n = S.nodes_per_level[l]; W->level_off[gl + 1] = n; if(n > W->max_nodes_per_level) W->max_nodes_per_level = n; }
    for(long long l = 0; l < L; l++) W->level_off[l + 1] += W->level_off[l];
    W->N = W->level_off[L];
    W->node_level.resize(W->N);
    for(long long l = 0; l < L; l++) for(long long n = W->level_off[l]; n < W->level_off[l + 1]; n++) W->node_level[n] = (int)l;
    long long E = 0; for(auto& S : W->segs) E += (long long)S.e_level.size();
    W->E = E; W->edge_from.resize(E); W->edge_to.resize(E); W->edge_label.resize(E);
    { long long e = 0; for(auto& S : W->segs) for(size_t k = 0; k < S.e_level.size(); k++, e++) {
          long long gl = S.level0 + S.e_level[k];
          W->edge_from[e] = (int)(W->level_off[gl] + S.e_from[k]); W->edge_to[e] = (int)(W->level_off[gl + 1] + S.e_to[k]); W->edge_label[e] = S.e_label[k]; } }
    // ---- contigs: the backbone haplotypes end to end (row h of every segment), then allele rows of the windows
    for(int h = 0; h < P.n_backbone; h++) {
        Contig C;
        for(auto& S : W->segs) { const u8* row = S.M.data() + (size_t)h * S.len; for(int i = 0; i < S.len; i++) if(row[i] != '_') { C.seq.push_back(row[i]); C.level.push_back((int)(S.level0 + i)); } }
        W->contigs.push_back(std::move(C));
    }
    for(int w = 0; w < P.n_windows; w++) {
        Segment& S = W->segs[W->win_seg[w]];
        int K = std::min(P.contigs_per_window, S.nh - P.n_backbone);
        for(int k = 0; k < K; k++) {
            int rowI = P.n_backbone + (int)(((long long)k * (S.nh - P.n_backbone)) / K);
            Contig C; C.window = w; C.row = rowI;
            const u8* row = S.M.data() + (size_t)rowI * S.len;
            for(int i = 0; i < S.len; i++) if(row[i] != '_') { C.seq.push_back(row[i]); C.level.push_back((int)(S.level0 + i)); }
            W->contigs.push_back(std::move(C));
        }
    }
    return W;
}

void gm_world_destroy(void* w) { delete (World*)w; }

// sizes[8]: L, N, E, n_contigs, total contig bases, n_windows, max nodes per level, n_segments
void gm_world_sizes(void* w, long long* s)
{
    World* W = (World*)w; long long tb = 0; for(auto& c : W->contigs) tb += (long long)c.seq.size();
    s[0] = W->L; s[1] = W->N; s[2] = W->E; s[3] = (long long)W->contigs.size(); s[4] = tb; s[5] = W->P.n_windows; s[6] = W->max_nodes_per_level; s[7] = (long long)W->segs.size();
}
void gm_world_graph(void* w, int* node_level, int* edge_from, int* edge_to, u8* edge_label)
{
    World* W = (World*)w;
    memcpy(node_level, W->node_level.data(), W->N * 4); memcpy(edge_from, W->edge_from.data(), W->E * 4);
    memcpy(edge_to, W->edge_to.data(), W->E * 4); memcpy(edge_label, W->edge_label.data(), W->E);
}
void gm_world_contigs(void* w, long long* off, u8* seq, int* level, int* window, int* row)
{
    World* W = (World*)w; long long o = 0; size_t i = 0;
    for(auto& c : W->contigs) { off[i] = o; memcpy(seq + o, c.seq.data(), c.seq.size()); memcpy(level + o, c.level.data(), c.level.size() * 4); window[i] = c.window; row[i] = c.row; o += (long long)c.seq.size(); i++; }
    off[i] = o;
}
// per window: first level, last level (of its last column), alleles, exon columns
void gm_world_windows(void* w, int* first_level, int* last_level, int* n_alleles, int* n_exon_cols)
{
    World* W = (World*)w;
    for(int k = 0; k < W->P.n_windows; k++) { Segment& S = W->segs[W->win_seg[k]]; first_level[k] = (int)S.level0; last_level[k] = (int)(S.level0 + S.len - 1); n_alleles[k] = S.nh;
        int ne = 0; for(u8 e : S.exon) ne += e; n_exon_cols[k] = ne; }
}
// aligned allele matrix of a window (rows x columns) and its exon mask
void gm_world_window_matrix(void* w, int k, u8* M, u8* exon)
{
    World* W = (World*)w; Segment& S = W->segs[W->win_seg[k]];
    if(M) memcpy(M, S.M.data(), S.M.size());
    if(exon) memcpy(exon, S.exon.data(), S.exon.size());
}
void gm_world_nodes_per_level(void* w, int* out /* [L] */)
{
    World* W = (World*)w; for(long long l = 0; l < W->L; l++) out[l] = (int)(W->level_off[l + 1] - W->level_off[l]);
}

void* gm_batch_create(void* w, const gm_batch_params* bp, const char* matrix_path)
{
    World* W = (World*)w;
    QualMatrix Q;
    if(!load_matrix(matrix_path, Q, g_err)) return nullptr;
    const int n = bp->n_pairs, RL = bp->read_len, nbk = W->P.n_backbone;
    Batch* B = new Batch(); B->n_pairs = n; B->read_len = RL;
    // per-position tables stretched to RL (readSimulator.cpp:193-205)
    std::vector<int> qpos(RL); std::vector<double> expNegLam(Q.len);
    for(int p = 0; p < Q.len; p++) expNegLam[p] = exp(-Q.indel[p]);
    for(int p = 0; p < RL; p++) { double fr = (double)p / (double)RL; int t = (int)floor(fr * (Q.len - 1) + 0.5); if(t < 0) t = 0; if(t >= Q.len) t = Q.len - 1; qpos[p] = t; }
    // window sampling weights (by length)
    std::vector<double> wcum; double wt = 0;
    for(int k = 0; k < W->P.n_windows; k++) { wt += W->segs[W->win_seg[k]].len; wcum.push_back(wt); }
    // contigs of every window (backbone contigs cover all)
    std::vector<std::vector<int>> wcontigs(W->P.n_windows);
    for(size_t c = 0; c < W->contigs.size(); c++) if(W->contigs[c].window >= 0) wcontigs[W->contigs[c].window].push_back((int)c);

    struct PairOut { std::vector<u8> b[2], q[2]; std::vector<int> tl[2]; std::vector<Aln> al[2]; int window; };
    const bool timing = getenv("GM_TIMING") != nullptr; double t0 = omp_get_wtime();
    std::vector<PairOut> outs(n);
#ifdef _OPENMP
    if(W->P.threads > 0) omp_set_num_threads(W->P.threads);
#endif
#pragma omp parallel for schedule(dynamic, 256)
    for(int p = 0; p < n; p++) {
        Rng r(bp->seed * 0x9E3779B97F4A7C15ull + (u64)p * 0xD1B54A32D192ED03ull + 12345);
        PairOut& O = outs[p]; O.window = -1;
        for(int attempt = 0; attempt < 50; attempt++) {
            int jump = (int)floor(bp->jump_mean + bp->jump_sd * r.normal() + 0.5); if(jump < 10) jump = 10;
            int F = jump + RL + 8;                  // fragment bases needed (a few spare for deletions)
            // ---- the fragment: bases + their levels
            std::vector<u8> fb; std::vector<int> fl; int srcContig = -1;
            const bool gene = W->P.n_windows > 0 && r.uni() < bp->frac_gene;
            int win = -1;
            if(gene) {
                double u = r.uni() * wt; win = (int)(std::lower_bound(wcum.begin(), wcum.end(), u) - wcum.begin()); if(win >= W->P.n_windows) win = W->P.n_windows - 1;
                Segment& S = W->segs[W->win_seg[win]];
                int row = r.below(S.nh);
                if(S.len < F + 40) continue;
                int c0 = r.below(S.len - F - 20);
                const u8* rowp = S.M.data() + (size_t)row * S.len;
                for(int i = c0; i < S.len && (int)fb.size() < F; i++) if(rowp[i] != '_') { fb.push_back(rowp[i]); fl.push_back((int)(S.level0 + i)); }
                if((int)fb.size() < F) continue;
                if(row < nbk) srcContig = row;
            } else {
                int h = r.below(nbk); const Contig& C = W->contigs[h];
                if((long long)C.seq.size() < F + 2) continue;
                long long s0 = r.below64((long long)C.seq.size() - F);
                fb.assign(C.seq.begin() + s0, C.seq.begin() + s0 + F); fl.assign(C.level.begin() + s0, C.level.begin() + s0 + F);
                srcContig = h;
                // which window (if any) does the fragment touch
                for(int k = 0; k < W->P.n_windows; k++) { Segment& S = W->segs[W->win_seg[k]]; if(fl.back() >= S.level0 && fl.front() < S.level0 + S.len) { win = k; break; } }
            }
            O.window = win;
            const bool flip = r.uni() < bp->p_flip;
            bool ok = true;
            for(int m = 0; m < 2 && ok; m++) {
                const bool upstream = (m == 0) != flip;
                const bool rev = !upstream;
                // sample the read in alignment (= reference) orientation; position-in-read for the quality model is mirrored on the reverse strand
                int idx = upstream ? 0 : jump;          // start-to-start jump between the mates
                std::vector<u8>& rb = O.b[m]; std::vector<u8>& rq = O.q[m]; std::vector<int>& tl = O.tl[m];
                rb.assign(RL, 0); rq.assign(RL, 0); tl.assign(RL, -1);
                auto sample = [&](int base, u8 under, int lvl) {
                    int pr_ = rev ? RL - 1 - base : base; int qp = qpos[pr_];
                    double u = r.uni(); const auto& fv = Q.freq[qp]; size_t k = 0; while(k + 1 < fv.size() && u > fv[k].second) k++;
                    u8 q = fv[k].first; bool errB = r.uni() < 1.0 - Q.correct[qp][q];
                    rb[base] = errB ? NUC[r.below(4)] : under; rq[base] = q; tl[base] = lvl;
                };
                for(int base = 0; base < RL; base++) {
                    int pr_ = rev ? RL - 1 - base : base; const double eL = expNegLam[qpos[pr_]];
                    int ins = r.poissonL(eL), del = r.poissonL(eL);
                    if(ins > 0) { for(int k = 0; k < ins && base < RL; k++) { sample(base, NUC[r.below(4)], -1); base++; } if(base >= RL) break; }
                    idx += del;
                    if(idx >= (int)fb.size()) { ok = false; break; }
                    sample(base, fb[idx], fl[idx]); idx++;
                }
                if(!ok) break;
                // an inserted base at either end is indistinguishable from a clipped one: make the ends real
                if(tl[0] < 0 || tl[RL - 1] < 0) { ok = false; break; }
                // ---- alignments
                int clipL = r.uni() < bp->p_no_clip ? 0 : (int)(r.beta14() * bp->clip_max);
                int clipR = r.uni() < bp->p_no_clip ? 0 : (int)(r.beta14() * bp->clip_max);
                std::vector<Aln>& AL = O.al[m]; AL.clear();
                std::vector<int> cand;
                if(srcContig >= 0) cand.push_back(srcContig);
                if(gene) {
                    // a handful of the window's contigs + backbone haplotypes, as bwa would report the closest sequences
                    const auto& wc = wcontigs[win]; int want = bp->gene_candidates;
                    for(int k = 0; k < want; k++) { int c = (r.uni() < 0.7 && !wc.empty()) ? wc[r.below((int)wc.size())] : r.below(nbk); if(std::find(cand.begin(), cand.end(), c) == cand.end()) cand.push_back(c); }
                } else if(r.uni() < bp->p_secondary) {
                    int ns = std::min(bp->max_secondary, r.geometric(0.5));
                    for(int k = 0; k < ns; k++) { int c = r.below(nbk); if(std::find(cand.begin(), cand.end(), c) == cand.end()) cand.push_back(c); }
                }
                for(size_t k = 0; k < cand.size(); k++) {
                    Aln a; a.contig = cand[k]; a.rev = rev; a.primary = false;
                    const Contig& C = W->contigs[cand[k]];
                    if(k > 0 && r.uni() < bp->p_random_secondary) {
                        // non-homologous placement: a plain match run somewhere else on the contig
                        int ml = RL - clipL - clipR; if((int)C.seq.size() < ml + 2) continue;
                        a.pos = (int)r.below64((long long)C.seq.size() - ml - 1); a.ops.clear();
                        if(clipL) a.ops.emplace_back(clipL, 'S');
                        a.ops.emplace_back(ml, 'M');
                        if(clipR) a.ops.emplace_back(clipR, 'S');
                        int sc = 0; for(int q = 0; q < ml; q++) sc += (C.seq[a.pos + q] == rb[clipL + q]) ? 1 : -4; a.as = sc;
                        if(r.uni() < bp->p_wrong_strand * 10) a.rev = !rev;
                        AL.push_back(a); continue;
                    }
                    if(!align_by_levels(C, rb.data(), tl.data(), RL, clipL, clipR, a)) continue;
                    AL.push_back(a);
                }
                if(AL.empty()) { ok = false; break; }
                // primary = best score (first of equals), then AS-descending (stable), processBAM.cpp:1945-1967
                size_t best = 0; for(size_t k = 1; k < AL.size(); k++) if(AL[k].as > AL[best].as && AL[k].rev == rev) best = k;
                if(AL[best].rev != rev) { ok = false; break; }
                AL[best].primary = true;
                std::stable_sort(AL.begin(), AL.end(), [](const Aln& x, const Aln& y) { return x.as > y.as; });
            }
            if(ok) break;
            if(attempt == 49) { O.b[0].clear(); }
        }
    }
    if(timing) fprintf(stderr, "gm_batch: simulate %.3f s\n", omp_get_wtime() - t0);
    // ---- assembly: sizes and offsets in pair order, then a parallel fill
    static const char OPS[] = "MIDNSHP=X";
    std::vector<long long> chOff(n + 1, 0), cgOff(n + 1, 0);
    for(int p = 0; p < n; p++) {
        PairOut& O = outs[p];
        if(O.b[0].empty() || O.al[0].empty() || O.al[1].empty()) { g_err = "read simulation failed for a pair (world too small for the fragment length?)"; delete B; return nullptr; }
        long long nc = 0, ng = 0;
        for(int m = 0; m < 2; m++) { nc += (long long)O.al[m].size(); for(auto& a : O.al[m]) ng += (long long)a.ops.size(); }
        chOff[p + 1] = chOff[p] + nc; cgOff[p + 1] = cgOff[p] + ng;
    }
    const long long NC = chOff[n], NG = cgOff[n];
    B->read_bases.resize((size_t)2 * n * RL); B->read_quals.resize((size_t)2 * n * RL); B->truth_level.resize((size_t)2 * n * RL);
    B->read_off.resize((size_t)2 * n + 1); B->chain_off.resize((size_t)2 * n + 1); B->read_primary.resize((size_t)2 * n); B->read_window.resize(n);
    B->chain_contig.resize(NC); B->chain_pos.resize(NC); B->chain_as.resize(NC); B->chain_reverse.resize(NC); B->cigar_off.resize(NC + 1); B->cigar.resize(NG);
    B->read_off[0] = 0; B->chain_off[0] = 0; B->cigar_off[0] = 0;
#pragma omp parallel for schedule(static)
    for(int p = 0; p < n; p++) {
        PairOut& O = outs[p];
        B->read_window[p] = O.window;
        long long c = chOff[p], g = cgOff[p];
        for(int m = 0; m < 2; m++) {
            const size_t r = (size_t)2 * p + m;
            memcpy(B->read_bases.data() + r * RL, O.b[m].data(), RL); memcpy(B->read_quals.data() + r * RL, O.q[m].data(), RL);
            memcpy(B->truth_level.data() + r * RL, O.tl[m].data(), (size_t)RL * 4);
            B->read_off[r + 1] = (int)((r + 1) * RL);
            for(auto& a : O.al[m]) {
                if(a.primary) B->read_primary[r] = (int)c;
                B->chain_contig[c] = a.contig; B->chain_pos[c] = a.pos; B->chain_as[c] = a.as; B->chain_reverse[c] = a.rev ? 1 : 0;
                for(auto& o : a.ops) { int code = (int)(strchr(OPS, o.second) - OPS); B->cigar[g++] = ((uint32_t)o.first << 4) | (uint32_t)code; }
                c++; B->cigar_off[c] = (int)g;
            }
            B->chain_off[r + 1] = (int)c;
        }
    }
    // release the per-pair buffers in parallel (millions of small blocks)
#pragma omp parallel for schedule(static)
    for(int p = 0; p < n; p++) { PairOut e; std::swap(outs[p], e); }
    if(timing) fprintf(stderr, "gm_batch: total %.3f s\n", omp_get_wtime() - t0);
    return B;
}
void gm_batch_destroy(void* b) { delete (Batch*)b; }
// sizes[4]: reads, bases, chains, cigar ops
void gm_batch_sizes(void* b, long long* s) { Batch* B = (Batch*)b; s[0] = 2LL * B->n_pairs; s[1] = (long long)B->read_bases.size(); s[2] = (long long)B->chain_contig.size(); s[3] = (long long)B->cigar.size(); }
void gm_batch_get(void* b, int* read_off, u8* bases, u8* quals, int* truth_level, int* read_window, int* chain_off, int* read_primary, int* chain_contig, int* chain_pos,
                  int* chain_as, u8* chain_reverse, int* cigar_off, uint32_t* cigar)
{
    Batch* B = (Batch*)b;
    memcpy(read_off, B->read_off.data(), B->read_off.size() * 4); memcpy(bases, B->read_bases.data(), B->read_bases.size()); memcpy(quals, B->read_quals.data(), B->read_quals.size());
    memcpy(truth_level, B->truth_level.data(), B->truth_level.size() * 4); memcpy(read_window, B->read_window.data(), B->read_window.size() * 4);
    memcpy(chain_off, B->chain_off.data(), B->chain_off.size() * 4); memcpy(read_primary, B->read_primary.data(), B->read_primary.size() * 4);
    memcpy(chain_contig, B->chain_contig.data(), B->chain_contig.size() * 4); memcpy(chain_pos, B->chain_pos.data(), B->chain_pos.size() * 4);
    memcpy(chain_as, B->chain_as.data(), B->chain_as.size() * 4); memcpy(chain_reverse, B->chain_reverse.data(), B->chain_reverse.size());
    memcpy(cigar_off, B->cigar_off.data(), B->cigar_off.size() * 4); memcpy(cigar, B->cigar.data(), B->cigar.size() * 4);
}

}  // extern "C"
