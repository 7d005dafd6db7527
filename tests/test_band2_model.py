"""CPU: the lane-by-lane model of the two-track band kernel (tools/band2/band2_model.cpp -- kernel_dp_band2.hip is its transliteration) beside every DP call of the oracle:
every call the model completes equals extensionAligner::fullNeedleman_diagonal_extension_gapJumper (mapper/aligner/extensionAligner.cpp:335-1556) in columns, score,
iterations, candidate cells and edges.  (The kernel itself, on the track arrays of the host flatten: tests/test_gpu_align.py::test_two_track_band_kernels_are_bit_exact.)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools", "band2"))
from tools import synth


def test_model_equals_the_oracle_on_gap_heavy_worlds():
    import run_model
    tot = np.zeros(32, np.int64)
    for wk, bk in ((dict(seed=52, G=12000, k=3, n_mut=8, mut_density=0.01, gap_frac=0.6), dict(seed=62, clip_max=48, p_no_clip=0.0)),
                   (dict(seed=51, G=12000, k=1, n_largegap=3, gap_frac=0.2), dict(seed=61))):
        w = synth.make_world(**wk)
        b = synth.make_batch(w, 600, **bk)
        o = run_model.run(w, b, threads=4)
        assert int(o[4]) == 0, "model differs from the oracle: %s" % o[13:21]
        tot += o
    assert tot[3] > 500 and tot[22] > 50 and tot[23] > 50 and tot[25] > 10          # completed calls; with early cells; met again; diff rule through a stored pointer


def test_model_equals_the_oracle_on_graph_m():
    import run_model
    w = synth.make_world_m(seed=2, n_levels=200000, n_windows=4)
    b = synth.make_batch_m(w, 1500, seed=1000, frac_gene=0.1)
    o = run_model.run(w, b, threads=4)
    assert int(o[4]) == 0 and int(o[3]) > 1500
