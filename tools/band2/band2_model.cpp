// band2_model.cpp -- CPU model (lane by lane, iteration by iteration) of the TWO-TRACK band kernel of round 6 (hla-la_amd/csrc/kernel_dp_band2.hip), run on the
// arguments of every DP call of the CPU oracle and compared with the oracle's result.  TEST / DESIGN INFRASTRUCTURE: it includes the oracle's source (the product never
// does) and exists so that the algorithm of the kernel -- which cells a lane holds, what it reads from its neighbour, the closed forms that replace the reference's maps --
// can be proven bit-exact against extensionAligner::fullNeedleman_diagonal_extension_gapJumper (mapper/aligner/extensionAligner.cpp:335-1556) on millions of calls
// without a GPU.  The kernel is a transliteration of band2::Model::run.
//
// The class of calls: every level within reach holds one or two nodes ("tracks"), every node at most four edges in the call's direction, and at most ONE gap-path jump
// (Graph::computeGapEdgePaths, Graph/Graph.cpp:347-476; jump candidates extensionAligner.cpp:757-786) of at least four edges starts within reach.  That is the
// neighbourhood of the backbone's gap stretches: a base track and a '_' track side by side, the jump across.
//
//   MAIN band:  cell (i, j, z) = (levels walked, read bases consumed, track) is computed on iteration t = i + j by lane j.
//   EARLY band: the jump creates cell (b, j, zB) on iteration a + j + 1, Delta = (b - a) - 1 iterations before the main band gets there; everything that descends from
//               it is computed Delta iterations early as well: cell (i, j, z), i >= b, on iteration i + j - Delta -- by the same lane j, in a second set of registers.
//   MERGE:      when the main band reaches a cell the early band has kept, the reference finds it in `scores` (:951-979): per matrix the larger value stays
//               (strictly greater overwrites, and every overwritten entry resets the patience, :1043-1062).  The early values come back through a ring of the last
//               32 iterations; the `diff` rule (:1007-1041) of a cell that was met again reads the CURRENT values of its predecessor, which the lanes keep for the
//               cells of the last two iterations whether they are in the frontier or not.
#include "../../oracle/hlala_oracle.cpp"

namespace band2 {
using namespace orc;

constexpr int NEG = -30000, ABSENT = -20000;
constexpr int LANES = 64, RING = 32, MAXD = 320, MAXREACH = 224, TIES = 16;
enum { FAIL_NONE = 0, FAIL_INELIGIBLE = 1, FAIL_REACH = 2, FAIL_ITERS = 4, FAIL_TIES = 5, FAIL_BASES = 6 };

struct Step { int n[2]; unsigned char tz[2][4], lab[2][4]; int eid[2][4];
              unsigned long long w; };      // w: what the KERNEL stages per step -- 16 bits per (source track z', target track z) pair, pair (z', z) at bits 16 * (2 z' + z):
                                            //    bit 0 any edge, 1 an edge with a real label, 2 a '_' edge, 3 the '_' edge precedes the first real edge (CSR order), 4..8 which of A C G T N label
                                            //    the real edges, 9..11 (pairs (z', 0) only) the edges of source node z'
static inline int base_code(unsigned char c) { return c == 'A' ? 0 : (c == 'C' ? 1 : (c == 'G' ? 2 : (c == 'T' ? 3 : (c == 'N' ? 4 : 5)))); }
static unsigned long long step_word(const Step& s)
{
    unsigned long long w = 0;
    for(int zs = 0; zs < 2; zs++) for(int z = 0; z < 2; z++) {
        unsigned p = 0; int firstReal = -1, firstGap = -1;
        for(int k = 0; k < s.n[zs]; k++) if(s.tz[zs][k] == z) {
            p |= 1u;
            if(s.lab[zs][k] == '_') { p |= 4u; if(firstGap < 0) firstGap = k; }
            else { p |= 2u; if(firstReal < 0) firstReal = k; const int bc = base_code(s.lab[zs][k]); if(bc < 5) p |= 1u << (4 + bc); }
        }
        if(firstGap >= 0 && (firstReal < 0 || firstGap < firstReal)) p |= 8u;
        if(z == 0) p |= (unsigned)s.n[zs] << 9;
        w |= (unsigned long long)p << (16 * (2 * zs + z));
    }
    return w;
}
struct Window {
    int nNodes[MAXREACH + 2]; Step st[MAXREACH + 2];
    int reach = 0;
    bool haveJump = false; int ja = 0, jzA = 0, jb = 0, jzB = 0, jpath = -1, jlen = 0;
};

static bool label_ok(unsigned char c) { return c == 'A' || c == 'C' || c == 'G' || c == 'T' || c == 'N' || c == '_'; }

// the levels ahead of (x0, direction) as tracks and steps; reach = number of steps that can be represented
static void build_window(const Graph& g, int x0, bool fwd, int want, Window& W)
{
    W = Window();
    if(want > MAXREACH) want = MAXREACH;
    int reach = 0;
    auto level_of = [&](int i) { return fwd ? x0 + i : x0 - i; };
    W.nNodes[0] = (int)g.level_nodes[x0].size();
    if(W.nNodes[0] > 2) { W.reach = 0; return; }
    int jumpLimit = want + 1;
    for(int i = 0; i < want; i++) {
        const int lv = level_of(i), nx = level_of(i + 1);
        if(nx < 0 || nx >= g.L) break;
        const int nn = (int)g.level_nodes[nx].size();
        if(nn > 2) break;
        Step s; memset(&s, 0, sizeof(s));
        bool ok = true;
        for(int z = 0; z < W.nNodes[i] && ok; z++) {
            const int node = g.level_nodes[lv][z];
            const std::vector<int>& ed = fwd ? g.out_e[node] : g.in_e[node];
            if(ed.size() > 4 || ed.empty()) { ok = false; break; }
            s.n[z] = (int)ed.size();
            for(size_t k = 0; k < ed.size(); k++) {
                const int e = ed[k]; const int other = fwd ? g.eto[e] : g.efrom[e];
                if(!label_ok(g.elabel[e])) { ok = false; break; }
                s.tz[z][k] = (unsigned char)g.node_rank[other]; s.lab[z][k] = g.elabel[e]; s.eid[z][k] = e;
            }
            // gap-path jumps that start here (paths of one edge are no-ops: the '_' edge itself pushes the same value into the same cell a moment earlier, :738-752 / :757-786)
            const auto& tbl = fwd ? g.jump_fwd : g.jump_bwd;
            auto it = tbl.find(node);
            if(it != tbl.end()) {
                int nLong = 0, other = -1, path = -1;
                for(auto& t : it->second) if((int)g.paths[t.second].size() >= 2) { nLong++; other = t.first; path = t.second; }
                if(nLong > 0) {
                    const int len = nLong == 1 ? (int)g.paths[path].size() : 0;
                    if(nLong > 1 || W.haveJump || len < 4 || len - 1 > RING - 4) { if(i < jumpLimit) jumpLimit = i; }
                    else { W.haveJump = true; W.ja = i; W.jzA = z; W.jb = i + len; W.jzB = g.node_rank[other]; W.jpath = path; W.jlen = len; }
                }
            }
        }
        if(!ok) break;
        s.w = step_word(s);
        W.st[i + 1] = s; W.nNodes[i + 1] = nn;
        reach = i + 1;
    }
    if(jumpLimit < reach) reach = jumpLimit;
    if(W.haveJump && W.jb > reach) { if(W.ja < reach) reach = W.ja; if(W.ja >= reach) W.haveJump = false; }
    if(W.haveJump && W.ja >= reach) W.haveJump = false;
    W.reach = reach;
}

struct Cand { bool exists; int newD, dsel, GGv, gbit, SGv, ssrc, sext, sgap; };
// back-pointer record of one track of a cell: kept | useD << 1 | useG << 2 | useS << 3 | dsel << 4 | gbit << 7 | ssrc << 8 | sext << 9 | sgap << 10
static inline unsigned pack_bt(bool kept, bool useD, bool useG, bool useS, const Cand& c) { return (kept ? 1u : 0u) | (useD ? 2u : 0u) | (useG ? 4u : 0u) | (useS ? 8u : 0u) | ((unsigned)(c.dsel & 7) << 4) | ((unsigned)c.gbit << 7) | ((unsigned)c.ssrc << 8) | ((unsigned)c.sext << 9) | ((unsigned)c.sgap << 10); }

struct RingE { int t; bool kept; int D, G, S; unsigned bits; };

struct Result { int earlyKept = 0, remet = 0, overwritten = 0, diffViaPointer = 0, tracks2 = 0; int fail = 0; bool have = false; int score = INT_MIN, iters = 0; int sb = 0, se = -1; long long cells = 0, edges = 0;
                std::vector<int> levels, edges_used; std::string gchars, schars; };

struct Model {
    const Graph& g;
    explicit Model(const Graph& g_) : g(g_) {}

    Result run(const std::string& sequence, int y0, int x0, int z0, bool fwd, unsigned int seed, int wantReach) const
    {
        Result R;
        const int seqLen = (int)sequence.size();
        const int jmax = fwd ? seqLen - y0 : y0;
        if(jmax > LANES - 1 || jmax < 1) { R.fail = FAIL_BASES; return R; }
        static thread_local Window W;
        build_window(g, x0, fwd, wantReach, W);
        if(W.reach < 1) { R.fail = FAIL_INELIGIBLE; return R; }
        const int reach = W.reach;
        const bool haveJump = W.haveJump; const int Delta = haveJump ? W.jlen - 1 : 0;
        auto level_of = [&](int i) { return fwd ? x0 + i : x0 - i; };
        auto base_of = [&](int j) -> unsigned char { return (unsigned char)(fwd ? sequence[y0 + j - 1] : sequence[y0 - j]); };

        // lane state: [band][lane][track]
        static thread_local int D1[2][LANES][2], G1[2][LANES][2], S1[2][LANES][2], D2[2][LANES][2];
        static thread_local int PD1[LANES][2], PG1[LANES][2], PS1[LANES][2], PD2[LANES][2];
        static thread_local RingE ring[RING][LANES][2];
        static thread_local unsigned mainBT[MAXD + 2][LANES][2], earlyBT[MAXD + 2][LANES][2];
        for(int b = 0; b < 2; b++) for(int j = 0; j < LANES; j++) for(int z = 0; z < 2; z++) { D1[b][j][z] = G1[b][j][z] = S1[b][j][z] = D2[b][j][z] = NEG; }
        for(int j = 0; j < LANES; j++) for(int z = 0; z < 2; z++) { PD1[j][z] = PG1[j][z] = PS1[j][z] = PD2[j][z] = NEG; }
        for(int s = 0; s < RING; s++) for(int j = 0; j < LANES; j++) for(int z = 0; z < 2; z++) ring[s][j][z].t = -1;
        D1[0][0][z0] = 0; PD1[0][z0] = 0;                 // :495-519
        mainBT[0][0][0] = mainBT[0][0][1] = 0;

        int curMax = 0, lastInc = 0; int fpBand = 0, fpT = 0, fpJ = 0, fpZ = z0;      // currentMaxima_coordinates.front()
        int cBest = NEG, nTies = 0; int tieX[TIES], tieZ[TIES];
        const int diagonals = seqLen + g.L - 1;
        int itersRun = 0, fail = 0;
        int t = 1;
        static thread_local Cand cand[2][LANES][2];
        for(;; t++) {
            if(t > diagonals || t - lastInc > 40) break;                                    // :553
            bool anyLive = false;
            for(int b = 0; b < 2 && !anyLive; b++) for(int j = 0; j <= jmax && !anyLive; j++) for(int z = 0; z < 2; z++) if(D1[b][j][z] > ABSENT || D2[b][j][z] > ABSENT) { anyLive = true; break; }
            if(!anyLive) { itersRun = std::min(lastInc + 40, diagonals); break; }
            if(t > MAXD) { fail = FAIL_ITERS; break; }
            itersRun = t;
            // ---- candidates of every cell of this iteration, from the state of the last two
            for(int b = 0; b < 2; b++) for(int j = 0; j <= jmax; j++) for(int z = 0; z < 2; z++) {
                Cand& c = cand[b][j][z]; c = Cand{false, NEG, -1, NEG, 0, NEG, 0, 0, 0};
                const int i = t - j + (b ? Delta : 0);
                if(i < 0 || i > reach) continue;
                if(b == 1 && (!haveJump || i < W.jb)) continue;
                int best = NEG, dsel = -1;
                // (everything of the step comes out of its 64-bit word, as in the kernel: the pair (source track zs, target track z) in O(1))
                const unsigned long long sw = i >= 1 ? W.st[i].w : 0ull;
                const unsigned pr[2] = {(unsigned)((sw >> (16 * (0 + z))) & 0xFFFFu), (unsigned)((sw >> (16 * (2 + z))) & 0xFFFFu)};
                const int deg[2] = {(int)((sw >> 9) & 7u), (int)((sw >> (32 + 9)) & 7u)};
                // from the m-2 diagonal: match / mismatch through every edge, sources in map order, edges in CSR order (:565-607): +2 when some real edge of the pair carries
                // the read base, else -5 (a '_' edge is a mismatch like any other)
                if(sw && j >= 1) {
                    const int bc = base_code(base_of(j));
                    for(int zs = 0; zs < 2; zs++) {
                        const int src = D2[b][j - 1][zs];
                        if(src <= ABSENT) continue;
                        if(z == 0) R.edges += deg[zs];
                        if(pr[zs] & 1u) { const int v = src + ((bc < 5 && ((pr[zs] >> (4 + bc)) & 1u)) ? 2 : -5); if(v > best) { best = v; dsel = zs; } }
                    }
                }
                // from the m-1 diagonal, D candidates: '_' edges (:738-752) and the gap-path jump (:757-786), sources in map order: the jump source has the lower level --
                // it comes first in a forward call, last in a backward one
                auto jumpD = [&]() {
                    if(b == 1 && haveJump && i == W.jb && z == W.jzB && (t - 1 - j) == W.ja) { const int J = D1[0][j][W.jzA]; if(J > ABSENT && J > best) { best = J; dsel = 4; } }
                };
                auto gapD = [&]() {
                    for(int zs = 0; zs < 2; zs++) {
                        const int src = D1[b][j][zs];
                        if(src <= ABSENT) continue;
                        if((pr[zs] & 4u) && src > best) { best = src; dsel = 2 + zs; }
                    }
                };
                if(fwd) { jumpD(); gapD(); } else { gapD(); jumpD(); }
                // gap in graph (:621-661): from (i, j - 1, z), open before extend
                if(j >= 1) {
                    const int sD = D1[b][j - 1][z], sG = G1[b][j - 1][z];
                    if(sD > ABSENT) { c.GGv = sD - 6; c.gbit = 0; if(sG > ABSENT && sG - 2 > c.GGv) { c.GGv = sG - 2; c.gbit = 1; } }
                }
                // gap in sequence (:664-754): from (i - 1, j, z'), per edge [open, extend]; a '_' edge opens nothing and extends for free.  Within a pair: open through the
                // first real edge, extend through the '_' edge when there is one (S + 0 beats S - 2), else through the first real edge; the '_' edge's extension comes before
                // the real edge's open only when it precedes it in CSR order
                for(int zs = 0; zs < 2; zs++) {
                    const int sD = D1[b][j][zs], sS = S1[b][j][zs];
                    if(sD <= ABSENT || !sw) continue;
                    if(z == 0) R.edges += deg[zs];
                    const unsigned p = pr[zs];
                    if(!(p & 1u)) continue;
                    const bool real = p & 2u, gap = p & 4u, gapFirst = p & 8u;
                    const int open = real ? sD - 6 : NEG;
                    const int extG = (gap && sS > ABSENT) ? sS : NEG, extR = (real && sS > ABSENT) ? sS - 2 : NEG;
                    if(gapFirst && extG > c.SGv) { c.SGv = extG; c.ssrc = zs; c.sext = 1; c.sgap = 1; }
                    if(open > c.SGv) { c.SGv = open; c.ssrc = zs; c.sext = 0; c.sgap = 0; }
                    if(extR > c.SGv) { c.SGv = extR; c.ssrc = zs; c.sext = 1; c.sgap = 0; }
                    if(!gapFirst && extG > c.SGv) { c.SGv = extG; c.ssrc = zs; c.sext = 1; c.sgap = 1; }
                }
                if(c.GGv > best) { best = c.GGv; dsel = 5; }                                 // :840-865
                if(c.SGv > best) { best = c.SGv; dsel = 6; }
                c.newD = best; c.dsel = dsel; c.exists = best > ABSENT;
                if(c.exists) R.cells++;
            }
            // ---- call maxima (:794-1073), filtering (:1076-1102)
            int mxStored = NEG, mxNew = NEG; bool anyOverwritten = false, equalNonZero = false;
            static thread_local int stD[2][LANES][2], stG[2][LANES][2], stS[2][LANES][2]; static thread_local bool kept[2][LANES][2];
            // first cell in map order that carries mxNew: map order = (level, read position, rank)
            long long firstKey = 0; int fB = 0, fJ = 0, fZ = 0; bool haveFirst = false;
            for(int b = 0; b < 2; b++) for(int j = 0; j <= jmax; j++) for(int z = 0; z < 2; z++) {
                const Cand& c = cand[b][j][z];
                kept[b][j][z] = c.exists && c.newD >= -16;                                   // :949
                const int i = t - j + (b ? Delta : 0);
                // what the early band left in `scores` for this cell of the main band
                const RingE* E = nullptr;
                if(b == 0 && haveJump && i >= W.jb && t - Delta >= 1) { const RingE& e = ring[(t - Delta) % RING][j][z]; if(e.t == t - Delta && e.kept) E = &e; }
                bool useD = true, useG = true, useS = true;
                if(kept[b][j][z]) {
                    int sD = c.newD, sG = c.GGv, sS = c.SGv;
                    bool overwritten = false;
                    if(b == 1) R.earlyKept++;
                    if(E) {                                                                // :951-979
                        R.remet++;
                        useD = c.newD > E->D; useG = c.GGv > E->G; useS = c.SGv > E->S;
                        overwritten = useD || useG || useS;
                        if(!useD) sD = E->D; if(!useG) sG = E->G; if(!useS) sS = E->S;
                    }
                    stD[b][j][z] = sD; stG[b][j][z] = sG; stS[b][j][z] = sS;
                    if(sD > mxStored) mxStored = sD;
                    if(overwritten) { anyOverwritten = true; R.overwritten++; }
                    // running maximum (:1043-1062)
                    const int x = level_of(i), y = fwd ? y0 + j : y0 - j;
                    const long long key = ((long long)x << 24) | ((long long)y << 4) | z;
                    if(c.newD > mxNew || (c.newD == mxNew && key < firstKey)) { mxNew = c.newD; firstKey = key; fB = b; fJ = j; fZ = z; haveFirst = true; }
                    if(c.newD == curMax) {
                        // the `diff` rule (:1007-1041): the step behind the STORED D pointer, the predecessor's CURRENT value
                        bool zero;
                        if(!E || useD) zero = (c.dsel == 2 || c.dsel == 3 || c.dsel == 4) || (c.dsel == 6 && (E && !useS ? (((E->bits >> 9) & 1) && ((E->bits >> 10) & 1)) : (c.sext && c.sgap)));
                        else {
                            // met again and not improved in D: the early band's pointer, followed through the merged GG / SG pointers of the same cell
                            R.diffViaPointer++;
                            const unsigned eb = E->bits; const int ds = (eb >> 4) & 7;
                            int prev = NEG;
                            if(ds == 0 || ds == 1) prev = j >= 1 ? PD2[j - 1][ds] : NEG;
                            else if(ds == 2 || ds == 3) prev = PD1[j][ds - 2];
                            else if(ds == 4) prev = E->D;
                            else if(ds == 5) { const bool gb = useG ? c.gbit : ((eb >> 7) & 1); prev = j >= 1 ? (gb ? PG1[j - 1][z] : PD1[j - 1][z]) : NEG; }
                            else { const int ss = useS ? c.ssrc : (int)((eb >> 8) & 1); const bool se = useS ? c.sext : ((eb >> 9) & 1); prev = se ? PS1[j][ss] : PD1[j][ss]; }
                            zero = (c.newD - prev) == 0;
                        }
                        if(!zero) equalNonZero = true;
                    }
                    (b ? earlyBT : mainBT)[t][j][z] = pack_bt(true, useD, useG, useS, c);
                } else {
                    (b ? earlyBT : mainBT)[t][j][z] = 0;
                    stD[b][j][z] = stG[b][j][z] = stS[b][j][z] = NEG;
                }
            }
            if(anyOverwritten) lastInc = t;
            if(mxNew > curMax) { curMax = mxNew; lastInc = t; fpBand = fB; fpT = t; fpJ = fJ; fpZ = fZ; }
            else if(equalNonZero) lastInc = t;
            (void)haveFirst;
            // ---- sequence-complete cells (:982-999): lane jmax
            for(int b = 0; b < 2; b++) for(int z = 0; z < 2; z++) if(kept[b][jmax][z]) {
                const int i = t - jmax + (b ? Delta : 0);
                bool wasIn = false; int oldD = NEG;
                if(b == 0 && haveJump && i >= W.jb && t - Delta >= 1) { const RingE& e = ring[(t - Delta) % RING][jmax][z]; if(e.t == t - Delta && e.kept) { wasIn = true; oldD = e.D; } }
                const int sD = stD[b][jmax][z];
                if(wasIn && sD == oldD) continue;
                if(sD > cBest) { cBest = sD; nTies = 0; }
                if(sD == cBest) { if(nTies < TIES) { tieX[nTies] = level_of(i); tieZ[nTies] = z; } nTies++; }
            }
            // ---- the next state: what the main band's lanes remember of their cells (frontier or not), the ring, the frontiers
            for(int j = 0; j <= jmax; j++) for(int z = 0; z < 2; z++) {
                PD2[j][z] = PD1[j][z];
                const int i = t - j;
                int pD = NEG, pG = NEG, pS = NEG;
                if(kept[0][j][z]) { pD = stD[0][j][z]; pG = stG[0][j][z]; pS = stS[0][j][z]; }
                else if(haveJump && i >= W.jb && t - Delta >= 1) { const RingE& e = ring[(t - Delta) % RING][j][z]; if(e.t == t - Delta && e.kept) { pD = e.D; pG = e.G; pS = e.S; } }
                PD1[j][z] = pD; PG1[j][z] = pG; PS1[j][z] = pS;
            }
            for(int j = 0; j <= jmax; j++) for(int z = 0; z < 2; z++) {
                RingE& e = ring[t % RING][j][z];
                e.t = t; e.kept = kept[1][j][z]; e.D = stD[1][j][z]; e.G = stG[1][j][z]; e.S = stS[1][j][z]; e.bits = earlyBT[t][j][z];
            }
            for(int b = 0; b < 2; b++) for(int j = 0; j <= jmax; j++) for(int z = 0; z < 2; z++) {
                const bool survive = kept[b][j][z] && (mxStored - stD[b][j][z]) <= 15;
                D2[b][j][z] = D1[b][j][z];
                D1[b][j][z] = survive ? stD[b][j][z] : NEG; G1[b][j][z] = survive ? stG[b][j][z] : NEG; S1[b][j][z] = survive ? stS[b][j][z] : NEG;
                const int i = t - j + (b ? Delta : 0);
                if(survive && i >= reach) fail = FAIL_REACH;
            }
            if(fail) break;
        }
        R.iters = itersRun;
        if(fail) { R.fail = fail; return R; }
        if(nTies > TIES) { R.fail = FAIL_TIES; return R; }

        // ---- end cell (:1381-1517)
        int ex = 0, ez = 0, ej = 0; bool haveEnd = false; int endScore = 0;
        if(nTies > 0) {
            std::vector<std::pair<std::string, int>> keys;
            for(int k = 0; k < nTies; k++) keys.push_back({std::to_string(tieX[k]) + "/" + std::to_string(tieZ[k]), k});
            std::sort(keys.begin(), keys.end());
            unsigned int s = seed;
            const int sel = rand_r(&s) % nTies;
            const int k = keys[(size_t)sel].second;
            ex = tieX[k]; ez = tieZ[k]; ej = jmax; haveEnd = true; endScore = cBest;
        } else if(curMax > 0) {
            const int i = fpT - fpJ + (fpBand ? Delta : 0);
            ex = level_of(i); ez = fpZ; ej = fpJ; haveEnd = true; endScore = curMax;
        }
        if(!haveEnd) return R;
        // ---- backtrace (:1109-1354) through the merged pointers
        std::vector<int> recL, recE; std::string recG, recS;
        int ci = fwd ? ex - x0 : x0 - ex, cj = ej, cz = ez, cm = 0, guard = 0;
        while(!(ci == 0 && cj == 0)) {
            if(++guard > 4 * MAXD) { R.fail = 99; return R; }
            // the record of matrix cm of cell (ci, cj, cz): the main band's when it kept the cell and improved this matrix (or met nothing), else the early band's
            const int tm = ci + cj;
            unsigned mb = (tm >= 0 && tm <= itersRun && tm <= MAXD) ? mainBT[tm][cj][cz] : 0;
            unsigned eb = 0;
            if(haveJump && ci >= W.jb) { const int te = tm - Delta; if(te >= 1 && te <= itersRun) eb = earlyBT[te][cj][cz]; }
            auto pick = [&](int bit) -> unsigned { if((mb & 1u) && ((mb >> bit) & 1u)) return mb; if(eb & 1u) return eb; return mb; };
            const unsigned char sc = cj >= 1 ? base_of(cj) : (unsigned char)0;
            if(cm == 0) {
                const unsigned r = pick(1); const int ds = (r >> 4) & 7;
                if(!(r & 1u)) { R.fail = 98; return R; }
                if(ds == 0 || ds == 1) {
                    const Step& st = W.st[ci]; int k = -1;
                    for(int q = 0; q < st.n[ds]; q++) if(st.tz[ds][q] == cz && st.lab[ds][q] == sc) { k = q; break; }
                    if(k < 0) for(int q = 0; q < st.n[ds]; q++) if(st.tz[ds][q] == cz) { k = q; break; }
                    recG.push_back((char)st.lab[ds][k]); recL.push_back(fwd ? level_of(ci) - 1 : level_of(ci)); recS.push_back((char)sc); recE.push_back(st.eid[ds][k]);
                    ci -= 1; cj -= 1; cz = ds; cm = 0;
                } else if(ds == 2 || ds == 3) {
                    const Step& st = W.st[ci]; int k = -1; const int zs = ds - 2;
                    for(int q = 0; q < st.n[zs]; q++) if(st.tz[zs][q] == cz && st.lab[zs][q] == '_') { k = q; break; }
                    recG.push_back('_'); recL.push_back(fwd ? level_of(ci) - 1 : level_of(ci)); recS.push_back('_'); recE.push_back(st.eid[zs][k]);
                    ci -= 1; cz = zs; cm = 0;
                } else if(ds == 4) {
                    std::vector<int> edgePath = g.paths.at((size_t)W.jpath); std::vector<int> lv;
                    for(int e : edgePath) lv.push_back(g.node_level[g.efrom[e]]);
                    if(fwd) { std::reverse(lv.begin(), lv.end()); std::reverse(edgePath.begin(), edgePath.end()); }
                    recL.insert(recL.end(), lv.begin(), lv.end()); recG.append(edgePath.size(), '_'); recS.append(edgePath.size(), '_'); recE.insert(recE.end(), edgePath.begin(), edgePath.end());
                    ci = W.ja; cz = W.jzA; cm = 0;
                } else if(ds == 5) cm = 1;
                else cm = 2;
            } else if(cm == 1) {
                const unsigned r = pick(2);
                recG.push_back('_'); recL.push_back(-1); recS.push_back((char)sc); recE.push_back(-1);
                cj -= 1; cm = ((r >> 7) & 1u) ? 1 : 0;
            } else {
                const unsigned r = pick(3); const int zs = (r >> 8) & 1, se = (r >> 9) & 1, sg = (r >> 10) & 1;
                const Step& st = W.st[ci]; int k = -1;
                for(int q = 0; q < st.n[zs]; q++) if(st.tz[zs][q] == cz && ((st.lab[zs][q] == '_') == (sg != 0))) { k = q; break; }
                recG.push_back((char)st.lab[zs][k]); recL.push_back(fwd ? level_of(ci) - 1 : level_of(ci)); recS.push_back('_'); recE.push_back(st.eid[zs][k]);
                ci -= 1; cz = zs; cm = se ? 2 : 0;
            }
        }
        if(fwd) { std::reverse(recG.begin(), recG.end()); std::reverse(recL.begin(), recL.end()); std::reverse(recS.begin(), recS.end()); std::reverse(recE.begin(), recE.end()); }
        R.have = true; R.score = endScore; R.levels = recL; R.edges_used = recE; R.gchars = recG; R.schars = recS;
        if(fwd) { R.sb = y0; R.se = y0 + ej - 1; } else { R.sb = y0 - ej; R.se = y0 - 1; }
        return R;
    }
};

struct Stats { long long withEarly = 0, withRemet = 0, withOverwritten = 0, withDiffPtr = 0; long long calls = 0, bases_ok = 0, eligible = 0, done = 0, mismatch = 0, fail[8] = {0}, withJump = 0, iters_done = 0; long long firstBad[8] = {0}; };

struct Observer : Aligner::DpObserver {
    Stats st; int wantReach = 0; long long prevCells = 0, prevEdges = 0; long long badCounters = 0;
    void seen(const Aligner& A, const std::string& sequence, int start_sequence, int startLevel, int startZ, bool fwd, unsigned int seedBefore, const Aligner::Ext& r, long long oCells, long long oEdges) override
    {
        st.calls++;
        const int jmax = fwd ? (int)sequence.size() - start_sequence : start_sequence;
        if(jmax > LANES - 1) return;
        st.bases_ok++;
        Model M(*A.g);
        // the levels a call is taken to reach: the read bases left, the margin, the chains of sequence gaps that trail the best cells (kernel_dp_band.hip), twice over for the
        // iterations a patience reset adds
        Result m = M.run(sequence, start_sequence, startLevel, startZ, fwd, seedBefore, wantReach > 0 ? wantReach : MAXREACH);
        if(m.fail == FAIL_INELIGIBLE || m.fail == FAIL_BASES) { st.fail[m.fail & 7]++; return; }
        st.eligible++;
        if(m.fail) { st.fail[m.fail < 8 ? m.fail : 7]++; return; }
        st.done++; st.iters_done += m.iters; if(m.earlyKept) st.withEarly++; if(m.remet) st.withRemet++; if(m.overwritten) st.withOverwritten++; if(m.diffViaPointer) st.withDiffPtr++;
        bool same = (m.have == r.have) && (m.iters == r.iters);
        if(same && m.have) same = m.score == r.score && m.levels == r.chain.levels && m.edges_used == r.chain.edges && m.gchars == r.chain.graph_aligned && m.schars == r.chain.sequence_aligned && m.sb == r.chain.sequence_begin && m.se == r.chain.sequence_end;
        if(same && (m.cells != oCells || m.edges != oEdges)) { same = false; if(getenv("B2_VERBOSE") && badCounters++ < 3) fprintf(stderr, "band2 counters: cells %lld vs %lld, edges %lld vs %lld (x0 %d y0 %d fwd %d)\n", m.cells, oCells, m.edges, oEdges, startLevel, start_sequence, (int)fwd); }
        if(!same) {
            if(st.mismatch < 1) {
                st.firstBad[0] = startLevel; st.firstBad[1] = start_sequence; st.firstBad[2] = startZ; st.firstBad[3] = fwd; st.firstBad[4] = m.iters; st.firstBad[5] = r.iters; st.firstBad[6] = m.have ? m.score : -999; st.firstBad[7] = r.have ? r.score : -999;
                if(getenv("B2_VERBOSE")) {
                    fprintf(stderr, "band2 mismatch: x0 %d y0 %d z0 %d fwd %d seqLen %d | model have %d iters %d score %d cols %zu sb %d se %d | oracle have %d iters %d score %d cols %zu sb %d se %d\n", startLevel, start_sequence, startZ, (int)fwd, (int)sequence.size(),
                            (int)m.have, m.iters, m.score, m.levels.size(), m.sb, m.se, (int)r.have, r.iters, r.score, r.chain.levels.size(), r.chain.sequence_begin, r.chain.sequence_end);
                    auto dump = [](const char* nm, const std::vector<int>& lv, const std::string& gc, const std::string& sc2) { fprintf(stderr, " %s:", nm); for(size_t i = 0; i < lv.size(); i++) fprintf(stderr, " %d%c%c", lv[i], gc[i], sc2[i]); fprintf(stderr, "\n"); };
                    if(m.have) dump("model ", m.levels, m.gchars, m.schars);
                    if(r.have) dump("oracle", r.chain.levels, r.chain.graph_aligned, r.chain.sequence_aligned);
                }
            }
            st.mismatch++;
        }
    }
};
}  // namespace band2

// Every pair of the batch through the oracle (n_threads host threads), the model beside every DP call.  out[0..]: calls, calls with <= 63 bases, eligible (window of at
// least one step), completed by the model, mismatches, fail reasons [5..12], first mismatch [13..20], iterations of the completed calls [21].
extern "C" int b2_run(orc_handle* h, const hlala_batch_in* in, int n_threads, long long* out)
{
    using namespace orc;
    Processor& P = h->P;
    if(n_threads <= 0) n_threads = omp_get_max_threads();
    band2::Stats total; std::string firstErr;
#pragma omp parallel num_threads(n_threads)
    {
        band2::Observer ob; if(const char* e = getenv("B2_REACH")) ob.wantReach = atoi(e);
        Aligner::observer() = &ob;
#pragma omp for schedule(dynamic, 16)
        for(int p = 0; p < in->n_pairs; p++) {
            try { align_one_pair(P, in, p, nullptr, nullptr, nullptr, 0); }
            catch(std::exception& e) {
#pragma omp critical
                if(firstErr.empty()) firstErr = e.what();
            }
        }
        Aligner::observer() = nullptr;
#pragma omp critical
        {
            total.withEarly += ob.st.withEarly; total.withRemet += ob.st.withRemet; total.withOverwritten += ob.st.withOverwritten; total.withDiffPtr += ob.st.withDiffPtr;
            total.calls += ob.st.calls; total.bases_ok += ob.st.bases_ok; total.eligible += ob.st.eligible; total.done += ob.st.done; total.iters_done += ob.st.iters_done;
            if(total.mismatch == 0 && ob.st.mismatch) for(int i = 0; i < 8; i++) total.firstBad[i] = ob.st.firstBad[i];
            total.mismatch += ob.st.mismatch; for(int i = 0; i < 8; i++) total.fail[i] += ob.st.fail[i];
        }
    }
    out[0] = total.calls; out[1] = total.bases_ok; out[2] = total.eligible; out[3] = total.done; out[4] = total.mismatch;
    for(int i = 0; i < 8; i++) out[5 + i] = total.fail[i];
    for(int i = 0; i < 8; i++) out[13 + i] = total.firstBad[i];
    out[21] = total.iters_done; out[22] = total.withEarly; out[23] = total.withRemet; out[24] = total.withOverwritten; out[25] = total.withDiffPtr;
    if(!firstErr.empty()) { g_err = firstErr; return -1; }
    return 0;
}
