#!/bin/bash
# round 4: how busy is the LDS array under the DP classes?  One PMC pass (kernel trace only): SQ_LDS_IDX_ACTIVE = all LDS-array cycles, SQ_LDS_BANK_CONFLICT = the extra
# cycles of conflicts, against SQ_BUSY_CYCLES and the instruction counts of the same launch
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out
B="--no-cpu-baseline --resident-only"
rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $R/gpurun_out/r04_lds -- python3 $R/bench.py --steps 1 --warmup 0 --single-batch $B > $R/gpurun_out/r04_lds.log 2>&1
echo "rc=$?"; tail -2 $R/gpurun_out/r04_lds.log | cut -c1-300
python3 - <<'PY'
import csv, glob, collections, os
f = glob.glob(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r04_lds/**/*counter_collection.csv", recursive=True)
if not f: print("no counter file"); raise SystemExit
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
seen = set()
for row in csv.DictReader(open(f[0])):
    k = row["Kernel_Name"].split("(")[0][:60]
    acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
    key = (k, row["Dispatch_Id"])
    if key not in seen: seen.add(key); n[k] += 1
for k, c in sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_BUSY_CYCLES", 0))[:12]:
    m = max(1, n[k])
    print("%-46s launches %d | " % (k, m) + "  ".join("%s %.3g" % (cn.replace("SQ_", ""), v / m) for cn, v in sorted(c.items())))
PY
