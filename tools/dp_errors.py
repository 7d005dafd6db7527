"""Which capacity do the flagged chains of a Graph M batch exceed?  (the kernel line of the failure is returned in hlala_chains_out.ll)"""
import collections, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as ge
from tools import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
P = ge.load_package()
w = synth.make_world_m(seed=2)
b = synth.make_batch_m(w, n, seed=77, frac_gene=1.0)
ctx = P.Context(w["graph"], w["contigs"], insert_mean=b["insert_mean"], insert_sd=b["insert_sd"], rng_seed=12345)
gb = ctx.batch(b); gb.align()
ch = gb.chains(1)
bad = np.nonzero(ch["status"] < 0)[0]
print("flagged chains", len(bad), "status", collections.Counter(ch["status"][bad].tolist()), "kernel line", collections.Counter(ch["ll"][bad].astype(int).tolist()))
src = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "hla-la_amd", "csrc", "kernel_dp.hip")).read().split("\n")
for ln in sorted(set(ch["ll"][bad].astype(int).tolist())):
    if 0 < ln <= len(src):
        print(ln, src[ln - 1].strip()[:160])
reads = np.searchsorted(b["chain_off"], bad, side="right") - 1
pairs = np.unique(reads // 2)
os.makedirs("gpurun_out", exist_ok=True)
np.save("gpurun_out/failing_pairs.npy", pairs)
print("pairs", pairs[:40].tolist())
