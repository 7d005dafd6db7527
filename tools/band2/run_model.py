"""Runs the CPU model of the two-track band kernel (tools/band2/band2_model.cpp) beside every DP call of the CPU oracle on Graph M batches and reports how many calls it
takes, how many it completes, and how many differ from the oracle (must be 0).
   python tools/band2/run_model.py [n_levels] [n_pairs] [frac_gene ...]"""
import sys, os, ctypes as C, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from tools import synth
from conftest import load_package
P = load_package()
so = os.path.join(ROOT, "tools", "_build", "libband2.so")
src = os.path.join(ROOT, "tools", "band2", "band2_model.cpp")
if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(src), os.path.getmtime(os.path.join(ROOT, "oracle", "hlala_oracle.cpp"))):
    os.makedirs(os.path.dirname(so), exist_ok=True)
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-Wno-sign-compare", "-fopenmp", "-shared", "-o", so, src])
L = C.CDLL(so)
L.orc_create.restype = C.c_void_p; L.orc_create.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
L.orc_last_error.restype = C.c_char_p
L.b2_run.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]


def run(w, b, threads=0):
    g, k1 = P.fill_struct(P.GraphDesc, w["graph"]); c, k2 = P.fill_struct(P.ContigsDesc, w["contigs"])
    prm = P.Params(); prm.insert_mean = b["insert_mean"]; prm.insert_sd = b["insert_sd"]; prm.rng_seed = 12345; prm.long_read_mode = 0; prm.max_columns = 384
    h = L.orc_create(C.byref(g), C.byref(c), C.byref(prm))
    assert h, L.orc_last_error()
    s, keep = P.fill_struct(P.BatchIn, b)
    out = np.zeros(32, np.int64)
    rc = L.b2_run(h, C.byref(s), threads, out.ctypes.data)
    assert rc == 0, L.orc_last_error()
    L.orc_destroy.argtypes = [C.c_void_p]; L.orc_destroy(h)
    return out


def report(tag, o):
    calls, okb, elig, done, bad = [int(x) for x in o[:5]]
    f = [int(x) for x in o[5:13]]
    print("%s: %d DP calls, %d with <= 63 bases, %d with a window (no window %d), completed %d (%.1f %% of all calls), fail-overs: reach %d, iterations %d, ties %d, other %d | MISMATCHES %d | iterations per completed call %.1f"
          % (tag, calls, okb, elig, f[1], done, 100.0 * done / max(1, calls), f[2], f[4], f[5], f[7], bad, o[21] / max(1, done)))
    print("   completed calls in which the early band kept a cell %d, the main band met such a cell again %d, overwrote one %d, evaluated the diff rule through a stored pointer %d" % tuple(int(x) for x in o[22:26]))
    if bad:
        print("   first mismatch: x0 %d y0 %d z0 %d fwd %d | model iters %d oracle iters %d | model score %d oracle score %d" % tuple(int(x) for x in o[13:21]))
    return bad


if __name__ == "__main__":
    nlev = int(sys.argv[1]) if len(sys.argv) > 1 else 300000
    npairs = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
    fgs = [float(x) for x in sys.argv[3:]] or [0.0, 0.3]
    w = synth.make_world_m(seed=2, n_levels=nlev)
    bad = 0
    for fg in fgs:
        b = synth.make_batch_m(w, npairs, seed=1000, frac_gene=fg)
        bad += report("frac_gene %.1f" % fg, run(w, b))
    sys.exit(1 if bad else 0)
