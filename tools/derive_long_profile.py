#!/usr/bin/env python3
"""gpurun_out/r05_long_{stats,sq,fetch,write} (tools/gpu_r5_long.sh) -> profiles/r05_long_kernel_stats.csv, profiles/r05_long_pmc.csv, profiles/r05_long_traffic.json
(HBM bytes per read and SQ counters of k_project_chains<ProjLdsLong>, the kernel of BASELINE config 5; what bench.py's long_reads.roofline reports as traffic).
FETCH_SIZE is doubled as /opt/skills/guides/MI355X_MICROARCH.md prescribes for gfx950; WRITE_SIZE is uncalibrated.  Run from the repo root."""
import collections, csv, glob, json, os, shutil, sys

sys.path.insert(0, os.getcwd())
import bench  # noqa: E402

TAG = sys.argv[1] if len(sys.argv) > 1 else "r05_long"
PMC_READS = int(sys.argv[2]) if len(sys.argv) > 2 else 20000          # reads of the counter passes (tools/gpu_r5_long.sh)


def newest(pat):
    f = sorted(glob.glob(pat), key=os.path.getmtime)
    return f[-1] if f else None


st = newest("gpurun_out/" + TAG + "_stats/*/*kernel_stats.csv")
shutil.copy(st, "profiles/" + TAG + "_kernel_stats.csv")
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for sub in ("sq", "fetch", "write"):
    f = newest("gpurun_out/%s_%s/*/*counter_collection.csv" % (TAG, sub))
    if not f:
        continue
    for r in csv.DictReader(open(f)):
        k = "k_project_chains<ProjLdsLong>" if "ProjLdsLong" in r["Kernel_Name"] else ("k_stitch_chains" if "k_stitch" in r["Kernel_Name"] else None)
        if k:
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
cn = ["SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "FETCH_SIZE", "WRITE_SIZE"]
with open("profiles/" + TAG + "_pmc.csv", "w") as o:
    o.write("kernel,reads_of_the_pass," + ",".join(c + "_per_read" for c in cn) + ",hbm_bytes_per_read_(2*FETCH+WRITE)*1024,wait_frac,active_frac\n")
    for k, v in acc.items():
        per = [v.get(c, 0.0) / PMC_READS for c in cn]
        hbm = (2 * v.get("FETCH_SIZE", 0.0) + v.get("WRITE_SIZE", 0.0)) * 1024 / PMC_READS
        o.write("%s,%d,%s,%.0f,%.4f,%.4f\n" % (k, PMC_READS, ",".join("%.4g" % x for x in per), hbm, v["SQ_WAIT_ANY"] / max(1.0, v["SQ_WAVE_CYCLES"]), v["SQ_ACTIVE_INST_ANY"] / max(1.0, v["SQ_WAVE_CYCLES"])))
v = acc["k_project_chains<ProjLdsLong>"]
hbm = (2 * v.get("FETCH_SIZE", 0.0) + v.get("WRITE_SIZE", 0.0)) * 1024 / PMC_READS
json.dump({"kernel": "k_project_chains<ProjLdsLong>", "kernel_source_hash": bench.kernel_source_hash(), "reads_of_the_counter_passes": PMC_READS, "hbm_bytes_per_read": hbm,
           "secondary": {"wait_frac": v["SQ_WAIT_ANY"] / max(1.0, v["SQ_WAVE_CYCLES"]), "active_frac": v["SQ_ACTIVE_INST_ANY"] / max(1.0, v["SQ_WAVE_CYCLES"]),
                         "valu_insts_per_read": v["SQ_INSTS_VALU"] / PMC_READS, "salu_insts_per_read": v["SQ_INSTS_SALU"] / PMC_READS, "lds_insts_per_read": v["SQ_INSTS_LDS"] / PMC_READS},
           "note": "(2*FETCH_SIZE + WRITE_SIZE)*1024 per read, separate --pmc passes of tools/long_profile.py; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950, WRITE_SIZE uncalibrated"},
          open("profiles/" + TAG + "_traffic.json", "w"), indent=1)
print("long reads: %.4g HBM bytes per read, wait %.2f" % (hbm, v["SQ_WAIT_ANY"] / max(1.0, v["SQ_WAVE_CYCLES"])))
