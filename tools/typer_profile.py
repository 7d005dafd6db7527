"""The HLATyper kernels at the size of real class-I loci: per-cluster x per-read likelihoods, all cluster pairs, the call.
usage: typer_profile.py [C R]...   (default 3000 400 and 5000 1000).  Host wall clock of the C-ABI calls incl. their transfers; run under
`rocprofv3 --kernel-trace --stats` for the kernel times (tools/gpu_typer_profile.sh -> profiles/r02_typer_kernel_stats.csv)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import __graft_entry__ as ge
from tools import synth
P = ge.load_package()
args = [int(x) for x in sys.argv[1:]] or [3000, 400, 5000, 1000]
w = synth.make_world(seed=1, G=2000, k=1)
ctx = P.Context(w["graph"], w["contigs"])
for Cn, R in zip(args[0::2], args[1::2]):
    loc = synth.make_locus(seed=5, n_clusters=Cn, exon_length=546, n_reads=R)
    ctx.exon_loglik(loc)
    t = time.perf_counter(); LL, M = ctx.exon_loglik(loc); t_e = time.perf_counter() - t
    ctx.pair_loglik(LL, M)
    t = time.perf_counter(); pl = ctx.pair_loglik(LL, M); t_p = time.perf_counter() - t
    ctx.call_locus(*pl)
    t = time.perf_counter(); call = ctx.call_locus(*pl); t_c = time.perf_counter() - t
    n = Cn * (Cn + 1) // 2 * R
    print("C=%d R=%d: exon_loglik %.2f ms, pair_loglik %.2f ms (%.1f G logAvg/s incl. transfers), call_locus %.2f ms; called clusters (%d, %d), simulated from %s"
          % (Cn, R, t_e * 1e3, t_p * 1e3, n / t_p / 1e9, t_c * 1e3, call["first_cluster"], call["second_cluster"], loc["truth"].tolist()))
