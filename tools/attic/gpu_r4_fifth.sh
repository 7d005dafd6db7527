#!/bin/bash
# round 4, fifth session: rethread with 4-byte back pointers (parity), JF switch fixed (A/B), JF at five waves per SIMD, CU mask with an own stream, TCC counters
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
bash tools/gpu_r4_ab.sh --parity "jf:HLALA_DP_JF=1" "nojf:HLALA_DP_JF=0"
bash tools/gpu_r4_ab.sh --modes two "own-stream:HLALA_BENCH_STREAM=own" "own-stream-sidecus64:HLALA_BENCH_STREAM=own HLALA_SIDE_CUS=64" "own-stream-sidecus128:HLALA_BENCH_STREAM=own HLALA_SIDE_CUS=128"
# TCC counters of the default build (one step, one batch)
cd /tmp && export TMPDIR=/tmp
B="--no-cpu-baseline --resident-only --steps 1 --warmup 0 --single-batch"
for p in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCC_HIT_sum TCC_MISS_sum TCC_READ_sum TCC_WRITE_sum" "TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_DRAM_sum TCC_NORMAL_WRITEBACK_sum TCC_NORMAL_EVICT_sum"; do
  d=$R/gpurun_out/r04q_tcc_$(echo $p | cut -d' ' -f1)
  rm -rf $d
  rocprofv3 --pmc $p --kernel-trace --output-format csv -d $d -- python3 $R/bench.py $B > $d.log 2>&1
  find $d -name "*kernel_trace.csv" -delete
done
cd $R && python3 - <<'PY'
import collections, csv, glob
acc = collections.defaultdict(lambda: collections.defaultdict(float)); nl = collections.defaultdict(set)
for f in glob.glob("gpurun_out/r04q_tcc_*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][-40:]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); nl[(k, r["Counter_Name"])].add(r["Dispatch_Id"])
names = sorted({c for k in acc for c in acc[k]})
print("kernel | " + " | ".join(n.replace("TCC_", "").replace("_sum", "") for n in names))
for k in sorted(acc, key=lambda k: -acc[k].get("TCC_EA0_RDREQ_sum", 0))[:12]:
    print(k, "|", " | ".join("%.3g" % (acc[k][n] / max(1, len(nl[(k, n)]))) for n in names))
PY
# JF at five waves per SIMD
rm -rf /tmp/w5 && mkdir -p /tmp/w5 && cp -r hla-la_amd include tools tests bench.py /tmp/w5/ && cd /tmp/w5
touch hla-la_amd/csrc/hlala_api.hip; make -C hla-la_amd/csrc EXTRA=-DHLALA_DP_TINYJF_WAVES=5 ../libhlala_gpu.so 2>&1 | grep -E "error"
GRAFT_REPO_ROOT=/tmp/w5 bash tools/gpu_r4_ab.sh "jf-5-waves:HLALA_X=1" | tee -a $R/gpurun_out/r4_ab.log
