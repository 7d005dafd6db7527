"""tools/gen_long.py <dir> <seed> <n_reads> <len_lo> <len_hi> <out.npz> -- one chunk of distinct long reads (synth.make_long_batch_fast) on the contigs
saved under <dir> (contig_off.npy, contig_seq.npy); bench.py and the tests start several of these side by side (numpy per read, one core each)."""
import os
import sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools import synth

d, seed, n, lo, hi, out = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), sys.argv[6]
off = np.load(os.path.join(d, "contig_off.npy")); seq = np.load(os.path.join(d, "contig_seq.npy"), mmap_mode="r")
starts = np.load(os.path.join(d, "starts_%d.npy" % seed)) if os.path.exists(os.path.join(d, "starts_%d.npy" % seed)) else None
w = {"contigs": {"contig_off": off, "contig_seq": seq, "n_contigs": len(off) - 1}}
b = synth.make_long_batch_fast(w, n, seed=seed, len_lo=lo, len_hi=hi, starts=starts)
np.savez(out, **{k: np.asarray(v) for k, v in b.items()})
