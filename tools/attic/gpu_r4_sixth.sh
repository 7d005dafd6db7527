#!/bin/bash
# round 4, sixth session: current build (jump-free switch fixed, 4-byte back pointers in k_rethread_chains) parity + A/B; non-temporal cell stores A/B + traffic
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
bash tools/gpu_r4_ab.sh --parity "jf:HLALA_DP_JF=1" "nojf:HLALA_DP_JF=0"
rm -rf /tmp/nt && mkdir -p /tmp/nt && cp -r hla-la_amd include tools tests bench.py /tmp/nt/ && cd /tmp/nt
touch hla-la_amd/csrc/hlala_api.hip; make -C hla-la_amd/csrc EXTRA=-DHLALA_DP_CELL_NT ../libhlala_gpu.so 2>&1 | grep -E "error"
GRAFT_REPO_ROOT=/tmp/nt bash tools/gpu_r4_ab.sh "nt-cell-stores:HLALA_X=1" | tee -a $R/gpurun_out/r4_ab.log
cd /tmp && export TMPDIR=/tmp
B="--no-cpu-baseline --resident-only --steps 1 --warmup 0 --single-batch"
for v in nt def; do
  src=$R; [ $v = nt ] && src=/tmp/nt
  for p in FETCH_SIZE WRITE_SIZE; do
    d=$R/gpurun_out/r04s_${v}_$p; rm -rf $d
    rocprofv3 --pmc $p --kernel-trace --output-format csv -d $d -- python3 $src/bench.py $B > $d.log 2>&1
    find $d -name "*kernel_trace.csv" -delete
  done
done
cd $R && python3 - <<'PY'
import collections, csv, glob
def counters(d, cn):
    acc = collections.defaultdict(float); n = collections.defaultdict(set)
    for f in glob.glob("gpurun_out/%s/*/*counter_collection.csv" % d):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == cn:
                k = r["Kernel_Name"].split("(")[0][-44:]
                acc[k] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
    return {k: acc[k] / len(n[k]) for k in acc}
for v in ("def", "nt"):
    f = counters("r04s_%s_FETCH_SIZE" % v, "FETCH_SIZE"); w = counters("r04s_%s_WRITE_SIZE" % v, "WRITE_SIZE")
    print("==", v)
    for k in sorted(f, key=lambda k: -(2 * f[k] + w.get(k, 0)))[:9]:
        print("%-46s fetch(x2) %7.2f GB  write %7.2f GB  total %7.2f GB" % (k, 2 * f[k] * 1024 / 1e9, w.get(k, 0) * 1024 / 1e9, (2 * f[k] + w.get(k, 0)) * 1024 / 1e9))
PY
