import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def load_golden(name="r01_small.npz"):
    z = np.load(os.path.join(HERE, "golden", name))
    g, c, b, e = {}, {}, {}, {}
    for k in z.files:
        sec, key = k.split("__", 1)
        v = z[k]
        if v.shape == ():
            v = v.item()
        {"graph": g, "contigs": c, "batch": b, "exp": e}[sec][key] = v
    return g, c, b, e


def check_against_golden(ext, pairs, e, stride):
    assert np.array_equal(ext["status"], e["ext_status"]) and np.array_equal(ext["n_cols"], e["ext_ncols"])
    ok2 = np.repeat(e["ext_status"] == 0, 2); ok = e["ext_status"] == 0      # skipped chains carry no DP results
    assert np.array_equal(ext["dp_iters"][ok2], e["dp_iters"][ok2]) and np.array_equal(ext["dp_score"][ok2], e["dp_score"][ok2])
    assert np.allclose(ext["ll"][ok], e["ext_ll"][ok], rtol=1e-12, atol=0)
    n = ext["n_cols"]; mask = np.arange(stride)[None, :] < n[:, None]
    assert np.array_equal((ext["col_level"].reshape(-1, stride).astype(np.int64) * mask).sum(1), e["ext_level_sum"])
    assert np.array_equal((ext["col_edge"].reshape(-1, stride).astype(np.int64) * mask).sum(1), e["ext_edge_sum"])
    for k in ("pair_status", "best_chain", "n_combinations", "strands_valid", "n_cols", "col_level", "col_gchar", "col_schar", "col_mapq"):
        assert np.array_equal(pairs[k], e[k]), k
    assert np.allclose(pairs["pair_ll"], e["pair_ll"], rtol=1e-12, atol=0)
    assert np.allclose(pairs["pair_mapq"], e["pair_mapq"], rtol=1e-9, atol=1e-15)
    assert np.allclose(pairs["mate_mapq"], e["mate_mapq"], rtol=1e-9, atol=1e-15)
