#!/usr/bin/env python3
"""Turn the two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of tools/gpu_bench_profile.sh into
profiles/r01_pmc_hbm_1Mpairs.csv and profiles/r01_traffic.json (HBM bytes per launch of the dominant kernel).
FETCH_SIZE is doubled as /opt/skills/guides/MI355X_MICROARCH.md prescribes for gfx950; WRITE_SIZE is uncalibrated."""
import collections, csv, glob, json, sys

fetch_dir, write_dir, launches = sys.argv[1], sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 1


def label(k):
    for pat, name in (("DpTiny", "k_dp<DpTiny,0>"), ("DpMid", "k_dp<DpMid,1>"), ("DpSmall", "k_dp<DpSmall,2>"), ("DpLarge", "k_dp<DpLarge,3>"),
                      ("k_stitch", "k_stitch_chains"), ("k_project", "k_project_chains"), ("k_pair_chains", "k_pair_chains"), ("k_dp_items", "k_dp_items")):
        if pat in k:
            return name
    return None


def summ(pat, cname):
    acc = collections.defaultdict(float)
    for f in glob.glob(pat):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == cname and label(r["Kernel_Name"]):
                acc[label(r["Kernel_Name"])] += float(r["Counter_Value"])
    return acc


f = summ(fetch_dir + "/*/*counter_collection.csv", "FETCH_SIZE")
w = summ(write_dir + "/*/*counter_collection.csv", "WRITE_SIZE")
rows = [(k, f[k] / launches, w.get(k, 0) / launches, (2 * f[k] + w.get(k, 0)) * 1024 / launches) for k in f]
with open("profiles/r01_pmc_hbm_1Mpairs.csv", "w") as o:
    o.write("kernel,FETCH_SIZE_KB_per_launch,WRITE_SIZE_KB_per_launch,hbm_bytes_per_launch_(2*FETCH+WRITE)*1024\n")
    for r in sorted(rows, key=lambda r: -r[3]):
        o.write("%s,%.3f,%.3f,%.0f\n" % r)
        print("%-18s %.4g bytes" % (r[0], r[3]))
t = [r for r in rows if r[0] == "k_dp<DpTiny,0>"][0]
json.dump({"pairs": 1048576, "levels": 5000000, "kernel": "k_dp<DpTiny, 0>", "fetch_size_kb": t[1], "write_size_kb": t[2],
           "hbm_bytes_per_launch": t[3],
           "note": "(2*FETCH_SIZE + WRITE_SIZE)*1024 per launch, separate --pmc passes of `bench.py --steps 1 --warmup 0`; FETCH_SIZE doubled "
                   "as MI355X_MICROARCH.md prescribes for gfx950, WRITE_SIZE uncalibrated"}, open("profiles/r01_traffic.json", "w"), indent=1)
