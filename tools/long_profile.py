"""BASELINE config 5 alone, for profilers: the long-read leg of bench.py (50 000 distinct reads of ~10 kb on the bench's Graph M, batches of 10 000 through a context with
16 384-column rows) without the rest of the bench.   python tools/long_profile.py [reads] [levels]"""
import os, sys, time, argparse
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import bench
from tools import synth
P = bench.load_package()
a = argparse.Namespace(long_reads=int(sys.argv[1]) if len(sys.argv) > 1 else 50000, long_reads_check=int(sys.argv[3]) if len(sys.argv) > 3 else 0, long_reads_batch=int(os.environ.get("LONG_BATCH", "50000")))
w = synth.make_world_m(seed=2, n_levels=int(sys.argv[2]) if len(sys.argv) > 2 else 5_000_000)
r = bench.long_reads(a, P, synth, w)
print("long reads: %d reads, %.0f reads/s, %.1f Mbases/s; stage ms (sum over %d batches): %s; ok %d, parity checked %s" % (r["reads"], r["reads_per_s"], r["bases_per_s"] / 1e6, r["batches"], r["stage_ms_sum"], r["reads_ok"], r["parity_checked"]))
