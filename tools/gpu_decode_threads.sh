#!/bin/bash
# how the BAM decoder's phases scale with the number of host threads on the box (HLALA_BAM_DEBUG clocks), and what first-touch of big buffers costs there
set -u
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
nproc; free -g | head -2
g++ -O2 -pthread -o /tmp/pagefault tools/micro/pagefault.cpp 2>/dev/null && /tmp/pagefault 12 | tee gpurun_out/r4_pagefault.txt
make -s -C tools/graphm 2>&1 | tail -1
python - <<'PY' 2>&1 | tee gpurun_out/r4_decode_threads.txt
import sys, time, ctypes as C, numpy as np, os
R = os.environ["GRAFT_REPO_ROOT"]
sys.path.insert(0, R); sys.path.insert(0, R + "/tests")
from conftest import load_package
from tools import synth
pkg = load_package()
w = synth.make_world_m(seed=2)
nct = w["contigs"]["n_contigs"]; clen = np.diff(w["contigs"]["contig_off"])
refs = [("ctg%d" % i, int(clen[i])) for i in range(nct)]
intervals = [("ctg%d" % i, 0, int(clen[i]) - 1, i) for i in range(nct)]
path = "/tmp/ab.bam"
bw = synth.BamWriter(path, refs, threads=0, level=1)
for k in range(8):
    b = synth.make_batch_m(w, 1 << 20, seed=3000 + k, frac_gene=0.04)
    names, rank = synth.scrambled_names(k, 1 << 20)
    bw.append_batch(b, names, order="coordinate"); del b
print("bam bytes", bw.close(), flush=True)
lib = C.CDLL(R + "/hla-la_amd/libhlala_host.so")
os.environ["HLALA_BAM_DEBUG"] = "1"
for T in (128, 64, 32, 16, 8, 32, 128):
    t = time.time(); S = pkg.bam_open_seeds(lib, path, intervals, threads=T); dt = time.time() - t
    tm = S.timing(); n = S.n_units
    t = time.time(); S.close(); tf = time.time() - t
    print("threads %3d: %.2f s (%.2f M pairs/s) free %.2f s | %s" % (T, dt, n / dt / 1e6, tf, {k: round(v, 2) for k, v in tm.items()}), flush=True)
PY
