"""The CPU model of the two-track band kernel on the random worlds of the parity sweep (tools/parity_sweep.py: stand-in graphs with random haplotype / mutation / gap
parameters, small Graph M worlds): every DP call the model completes must equal the oracle's.   python tools/band2/sweep_model.py [worlds] [pairs] [first seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, os.path.join(ROOT, 'tools', 'band2'))
import numpy as np
from tools import synth
import run_model


def world(s, n):
    rng = np.random.default_rng(s)
    read_len = int(rng.choice([76, 100, 125, 150, 151, 250]))
    if s % 3 == 2:
        al_lo = int(rng.choice([50, 400, 1500])); al_hi = al_lo + int(rng.choice([100, 1000, 3000])); nwin = int(rng.integers(1, 4))
        w = synth.make_world_m(seed=s, n_levels=int(rng.integers(30_000, 90_000)), n_windows=nwin, alleles=(al_lo, al_hi), n_backbone=int(rng.integers(2, 9)),
                               backbone_div=float(rng.choice([0.001, 0.003, 0.01])), gap_stretch_frac=float(rng.choice([0.0, 0.02, 0.08])))
        fg = float(rng.choice([0.1, 0.5, 1.0]))
        b = synth.make_batch_m(w, n, seed=s + 1, read_len=read_len, jump_mean=float(read_len + rng.integers(120, 300)), jump_sd=float(rng.integers(15, 60)), clip_max=int(rng.integers(0, read_len // 2)),
                               frac_gene=fg, p_secondary=float(rng.random()), max_secondary=int(rng.integers(1, 7)), p_random_secondary=float(rng.random() * 0.3))
        return w, b, "graph M"
    k = int(rng.choice([0, 1, 2, 3, 5])); G = int(rng.integers(8_000, 45_000))
    kw = dict(n_mut=int(rng.integers(2, 9)))
    if rng.random() < 0.4: kw["mut_density"] = float(rng.choice([0.01, 0.04, 0.08]))
    if rng.random() < 0.4: kw["n_largegap"] = int(rng.integers(1, 4))
    if rng.random() < 0.3: kw["gap_frac"] = float(rng.choice([0.2, 0.6]))
    if rng.random() < 0.3: kw["extra_identical"] = int(rng.integers(1, 3))
    w = synth.make_world(seed=s, G=G, k=k, **kw)
    read_len = max(read_len, 100)
    b = synth.make_batch(w, n, seed=s + 1, read_len=read_len, ins_mean=float(read_len + rng.integers(20, 200)), ins_sd=float(rng.integers(10, 70)), clip_max=int(rng.integers(0, read_len // 2)),
                         p_secondary=float(rng.random()), max_secondary=int(rng.integers(1, 7)), p_random_secondary=float(rng.random() * 0.4), indel_read_frac=float(rng.choice([0.0, 0.05, 0.3])), p_no_clip=float(rng.random() * 0.3))
    return w, b, "stand-in k %d %s" % (k, kw)


if __name__ == "__main__":
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 1500
    s0 = int(sys.argv[3]) if len(sys.argv) > 3 else 9000
    bad = 0; tot = np.zeros(32, np.int64)
    for s in range(s0, s0 + N):
        w, b, what = world(s, n)
        try:
            o = run_model.run(w, b)
        except AssertionError as e:
            print("seed %d skipped (%s): %s" % (s, what, str(e)[:100])); continue
        bad += run_model.report("seed %d (%s)" % (s, what), o); tot[:13] += o[:13]; tot[21:26] += o[21:26]
    run_model.report("ALL", tot)
    print("BAND2 MODEL SWEEP", "OK" if not bad else "FAILED: %d mismatches" % bad)
    sys.exit(1 if bad else 0)
