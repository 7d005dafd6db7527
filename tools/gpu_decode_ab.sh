#!/bin/bash
# A/B of BAM decoder builds on one box: A = the tree at git ref $1 (host_bam.cpp taken from gpurun_in_host_bam_A.cpp at the repo root), B = the tree's file
set -u
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out /tmp/a
cp -r hla-la_amd include tools tests /tmp/a/ && cp gpurun_in_host_bam_A.cpp /tmp/a/hla-la_amd/csrc/host_bam.cpp
( cd /tmp/a/hla-la_amd/csrc && g++ -O2 -std=c++17 -fPIC -shared -pthread -o ../libhlala_host.so host_check.cpp flat_graph.cpp host_filters.cpp host_loaders.cpp host_bam.cpp host_typer.cpp -lz 2>&1 | grep -E "rror" )
python - <<'PY'
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
for k in range(6):
    b = synth.make_batch_m(w, 1 << 20, seed=1000 + k, frac_gene=0.04)
    names, rank = synth.scrambled_names(k, 1 << 20)
    bw.append_batch(b, names, order="coordinate"); del b
print("bam bytes", bw.close(), flush=True)
for rep in range(3):
    for label, so, env in (("A", "/tmp/a/hla-la_amd/libhlala_host.so", None), ("B", R + "/hla-la_amd/libhlala_host.so", None), ("B-256MB", R + "/hla-la_amd/libhlala_host.so", "268435456")):
        if env: os.environ["HLALA_BAM_SEGMENT_BYTES"] = env
        else: os.environ.pop("HLALA_BAM_SEGMENT_BYTES", None)
        lib = C.CDLL(so)
        t = time.time(); S = pkg.bam_open_seeds(lib, path, intervals, threads=0); dt = time.time() - t
        tm = S.timing(); n = S.n_units
        t = time.time(); S.close(); tf = time.time() - t
        print("%-8s rep %d: %.2f s (%.2f M pairs/s) free %.2f s | %s" % (label, rep, dt, n / dt / 1e6, tf, {k: round(v, 2) for k, v in tm.items()}), flush=True)
PY
