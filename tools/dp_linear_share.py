"""Share of the extension-DP calls of the Graph M workload whose reach stays inside a LINEAR stretch of the graph (one node per level, one edge
between consecutive levels, label != '_', no gap-path jump) -- the calls the register-resident band kernel (kernel_dp_band.hip) can take.
From the CPU oracle's seed chains (stage A) and numpy over the graph arrays; no GPU.
   python tools/dp_linear_share.py [n_levels] [n_pairs]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from tools import synth
from oracle_binding import Oracle


def linear_runs(g, jf_levels=None, jb_levels=None):
    """lin_label[x] = label of the only edge between levels x and x + 1 when both hold one node (0 otherwise), and the run lengths
    lin_out[x] (levels x, x + 1, ... with a linear step ahead) / lin_in[x] (levels x, x - 1, ... with a linear step behind)."""
    L = g["n_levels"]; nl = g["node_level"]; ef = g["edge_from"]; el = g["edge_label"]
    npl = np.bincount(nl, minlength=L)
    elv = nl[ef]
    epl = np.bincount(elv, minlength=L)            # edges leaving level x
    lab = np.zeros(L, np.uint8)
    one = (npl[:-1] == 1) & (npl[1:] == 1) & (epl[:-1] == 1)
    first = np.zeros(L, np.int64); first[elv] = np.arange(len(ef))     # (any edge of the level; exact where there is one)
    lab[:-1][one] = el[first[:-1][one]]
    lab[lab == ord('_')] = 0
    return lab, npl


def run_lengths(ok_fwd, ok_bwd):
    L = len(ok_fwd)
    out = np.zeros(L, np.int32); inn = np.zeros(L, np.int32)
    run = 0
    for x in range(L - 1, -1, -1):
        run = min(255, run + 1) if ok_fwd[x] else 0
        out[x] = run
    run = 0
    for x in range(L):
        run = min(255, run + 1) if ok_bwd[x] else 0
        inn[x] = run
    return out, inn


def main():
    nlev = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    npairs = int(sys.argv[2]) if len(sys.argv) > 2 else 6000
    w = synth.make_world_m(seed=2, n_levels=nlev)
    g = w["graph"]
    lab, npl = linear_runs(g)
    L = g["n_levels"]
    print("levels", L, "single-node levels %.3f" % (npl == 1).mean(), "linear steps %.3f" % (lab != 0).mean())
    # (gap-path jumps start at nodes with a '_' out-edge; levels with such a node are not single-edge-non-gap, so a linear step never has a jump at its source;
    #  the product's flatten checks the jump tables themselves)
    ok_f = lab != 0                                   # step x -> x + 1
    ok_b = np.zeros(L, bool); ok_b[1:] = lab[:-1] != 0   # step x -> x - 1
    lin_out, lin_in = run_lengths(ok_f, ok_b)
    for fg in (0.3, 0.0, 1.0):
        b = synth.make_batch_m(w, npairs, seed=1000, frac_gene=fg)
        o = Oracle(g, w["contigs"], insert_mean=b["insert_mean"], insert_sd=b["insert_sd"], rng_seed=12345, max_columns=384)
        r = o.align_batch(b, stop_after_projection=True)
        s = r["seeds"]
        st = s["status"]; nc = s["n_cols"]; sb = s["seq_begin"]; se = s["seq_end"]
        stride = 384
        edges = s["col_edge"].reshape(-1, stride)
        nl = g["node_level"]; ef = g["edge_from"]; et = g["edge_to"]
        read_of_chain = np.repeat(np.arange(len(b["chain_off"]) - 1), np.diff(b["chain_off"]))
        rlen = np.diff(b["read_off"])[read_of_chain]
        ok = (st == 0) & (nc > 0)
        tot = 0; hist = {}
        for margin in (8, 12, 16):
            q = 0; tot = 0
            for c in np.nonzero(ok)[0]:
                e0 = edges[c, 0]; e1 = edges[c, nc[c] - 1]
                if e0 < 0 or e1 < 0: continue
                if sb[c] > 0:
                    lvl = nl[ef[e0]]
                    if lvl > 0:
                        tot += 1; reach = sb[c] + margin
                        if sb[c] <= 64 and lin_in[lvl] >= reach: q += 1
                if se[c] < rlen[c] - 1:
                    lvl = nl[et[e1]]
                    if lvl < L - 1:
                        tot += 1; reach = rlen[c] - 1 - se[c] + margin
                        if rlen[c] - 1 - se[c] <= 64 and lin_out[lvl] >= reach: q += 1
            hist[margin] = q / max(1, tot)
        print("frac_gene %.1f: DP calls (before sharing) %d; inside a linear stretch with margin 8 / 12 / 16: %.3f / %.3f / %.3f" % (fg, tot, hist[8], hist[12], hist[16]))


if __name__ == "__main__":
    main()
