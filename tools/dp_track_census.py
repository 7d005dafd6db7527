"""Round 6 census (CPU): what the neighbourhood of every extension-DP call of the Graph M workload looks like, by the structures a register band kernel would have
to hold -- nodes per level within reach, '_' edges, gap-path jump sources met in the call's direction.  From the oracle's seed chains (stage A) + numpy.
   python tools/dp_track_census.py [n_levels] [n_pairs] [frac_gene]"""
import sys, os, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from tools import synth
from oracle_binding import Oracle


def main():
    nlev = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    npairs = int(sys.argv[2]) if len(sys.argv) > 2 else 6000
    fgs = [float(sys.argv[3])] if len(sys.argv) > 3 else [0.0, 0.3, 1.0]
    w = synth.make_world_m(seed=2, n_levels=nlev)
    g = w["graph"]; L = g["n_levels"]
    nl = g["node_level"]; ef = g["edge_from"]; et = g["edge_to"]; el = g["edge_label"]
    npl = np.bincount(nl, minlength=L)
    gap = el == ord('_')
    # stretch starts (nodes with a '_' out-edge that are not themselves entered through a '_' edge) and multi-edge paths: level of the jump source / target
    has_gap_out = np.zeros(g["n_nodes"], bool); has_gap_out[ef[gap]] = True
    has_gap_in = np.zeros(g["n_nodes"], bool); has_gap_in[et[gap]] = True
    has_real_out = np.zeros(g["n_nodes"], bool); has_real_out[ef[~gap]] = True
    src_nodes = has_gap_out & ~has_gap_in                          # forward jump sources (paths of length >= 1 start here)
    # a path of ONE edge: source's '_' edge reaches a node with a real out-edge.  Multi-edge when the '_' target has a '_' out-edge.
    tgt_cont = np.zeros(g["n_nodes"], bool)
    e_gap = np.nonzero(gap)[0]
    tgt_cont[ef[e_gap]] |= has_gap_out[et[e_gap]]                  # node whose '_' edge leads to a node that continues with '_'
    jsrc_f = src_nodes & tgt_cont                                  # forward jump source with a path of >= 2 edges
    end_nodes = has_gap_in & has_real_out                          # path ends (backward jump sources)
    src_cont = np.zeros(g["n_nodes"], bool)
    src_cont[et[e_gap]] |= has_gap_in[ef[e_gap]]                   # node entered by a '_' edge whose source was itself entered by '_'
    jsrc_b = end_nodes & src_cont
    lvF = np.zeros(L, np.int32); np.add.at(lvF, nl[jsrc_f], 1)
    lvB = np.zeros(L, np.int32); np.add.at(lvB, nl[jsrc_b], 1)
    gapLevel = np.zeros(L, np.int32); np.add.at(gapLevel, nl[ef[gap]], 1)     # '_' edges leaving the level
    c_npl3 = np.concatenate([[0], np.cumsum(npl >= 3)]); c_npl2 = np.concatenate([[0], np.cumsum(npl == 2)])
    c_F = np.concatenate([[0], np.cumsum(lvF)]); c_B = np.concatenate([[0], np.cumsum(lvB)]); c_gap = np.concatenate([[0], np.cumsum(gapLevel > 0)])
    maxn = npl
    print("levels", L, "fwd jump sources (>= 2 edges)", int(jsrc_f.sum()), "bwd", int(jsrc_b.sum()), "levels with 2 nodes %.4f, >= 3 nodes %.4f" % ((npl == 2).mean(), (npl >= 3).mean()))
    for fg in fgs:
        b = synth.make_batch_m(w, npairs, seed=1000, frac_gene=fg)
        o = Oracle(g, w["contigs"], insert_mean=b["insert_mean"], insert_sd=b["insert_sd"], rng_seed=12345, max_columns=384)
        r = o.align_batch(b, stop_after_projection=True)
        s = r["seeds"]; st = s["status"]; nc = s["n_cols"]; sb = s["seq_begin"]; se = s["seq_end"]
        edges = s["col_edge"].reshape(-1, 384)
        read_of_chain = np.repeat(np.arange(len(b["chain_off"]) - 1), np.diff(b["chain_off"]))
        rlen = np.diff(b["read_off"])[read_of_chain]
        ok = (st == 0) & (nc > 0)
        cls = collections.Counter(); tot = 0
        basesHist = collections.Counter()
        for c in np.nonzero(ok)[0]:
            e0 = edges[c, 0]; e1 = edges[c, nc[c] - 1]
            if e0 < 0 or e1 < 0: continue
            for fwd in (False, True):
                if fwd:
                    bases = rlen[c] - 1 - se[c]
                    if bases <= 0: continue
                    x0 = nl[et[e1]]
                    if x0 >= L - 1: continue
                else:
                    bases = sb[c]
                    if bases <= 0: continue
                    x0 = nl[ef[e0]]
                    if x0 <= 0: continue
                tot += 1
                reach = int(bases + 8 + min(bases + 6, 40))
                lo, hi = (x0, min(L - 1, x0 + reach)) if fwd else (max(0, x0 - reach), x0)
                n3 = c_npl3[hi + 1] - c_npl3[lo]; n2 = c_npl2[hi + 1] - c_npl2[lo]
                # jump sources the call can stand on in its direction: forward sources in [x0, hi), backward sources in (lo, x0]
                nj = (c_F[hi] - c_F[lo]) if fwd else (c_B[hi + 1] - c_B[lo + 1])
                ng = c_gap[hi + 1] - c_gap[lo]
                big = bases > 48
                if n3: k = "3+ nodes on a level"
                elif n2 == 0 and ng == 0: k = "linear (band today, or parallel edges)"
                elif n2 == 0: k = "single nodes, parallel '_' edge"
                elif nj == 0 and ng == 0: k = "two tracks, no '_'"
                elif nj == 0: k = "two tracks with '_' edges, no jump source ahead"
                elif nj == 1: k = "two tracks, ONE jump source ahead"
                else: k = "two tracks, 2+ jump sources ahead"
                cls[(k, "bases > 48" if big else "bases <= 48")] += 1
                basesHist[min(int(bases) // 16, 9)] += 1
        print("frac_gene %.1f: %d DP calls (before sharing)" % (fg, tot))
        for (k, bg), v in sorted(cls.items(), key=lambda kv: -kv[1]):
            print("   %-52s %-12s %8d  %.3f" % (k, bg, v, v / tot))
        print("   bases left / 16 histogram:", sorted(basesHist.items()))


if __name__ == "__main__":
    main()
