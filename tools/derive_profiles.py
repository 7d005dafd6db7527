#!/usr/bin/env python3
"""gpurun_out/<tag>_{stats,fetch,write,sq} (tools/gpu_profile.sh <tag>) -> profiles/<tag>_* (tag = r02, r03, ...: `python tools/derive_profiles.py r03`):
  r02_kernel_stats_1Mpairs.csv   rocprofv3 --kernel-trace --stats summary (bench.py --steps 3 --warmup 1: 4 launches per kernel)
  r02_pmc_hbm_1Mpairs.csv        per kernel FETCH_SIZE / WRITE_SIZE (KB) and (2*FETCH_SIZE + WRITE_SIZE)*1024 bytes per launch -- FETCH_SIZE doubled
                                 for gfx950 as /opt/skills/guides/MI355X_MICROARCH.md prescribes, WRITE_SIZE uncalibrated; separate --pmc passes
  r02_pmc_sq_1Mpairs.csv         per kernel SQ counters of one launch (wave cycles, wait / active cycles, VALU / SALU / LDS instructions)
  r02_traffic.json               what bench.py reports as roofline.traffic / roofline.secondary for the dominant kernel, tagged with the hash
                                 of the kernel sources it was measured on (bench.py drops it when the sources change)
Run from the repo root AFTER the profile call, with the kernel sources unchanged."""
import collections, csv, glob, json, os, shutil, sys

sys.path.insert(0, os.getcwd())
import bench  # noqa: E402

TAG = sys.argv[1] if len(sys.argv) > 1 else "r02"

NAMES = (("k_dp_band<16>", "k_dp_band<16>"), ("k_dp_band<32>", "k_dp_band<32>"), ("k_dp_band<64>", "k_dp_band<64>"), ("DpTinyJF", "k_dp<DpTinyJF, 0>"), ("DpTiny", "k_dp<DpTiny, 0>"), ("DpMid", "k_dp<DpMid, 1>"), ("DpSmall", "k_dp<DpSmall, 2>"), ("DpWide", "k_dp<DpWide, 3>"), ("DpBroad", "k_dp<DpBroad, 4>"), ("DpLarge", "k_dp<DpLarge, 5>"), ("DpHuge", "k_dp<DpHuge, 6>"),
         ("k_stitch", "k_stitch_chains"), ("k_project", "k_project_chains"), ("k_rethread", "k_rethread_chains"), ("k_pair_chains", "k_pair_chains"), ("k_pair_multi", "k_pair_multi"), ("k_dp_band2", "k_dp_band2"), ("k_dp_items", "k_dp_items"), ("k_filter", "k_filter_chains"))


def label(k):
    for pat, name in NAMES:
        if pat in k:
            return name
    return None


def counters(d):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); launches = collections.defaultdict(set)
    files = sorted(glob.glob(os.path.join("gpurun_out", d, "*", "*counter_collection.csv")), key=os.path.getmtime)
    for f in files[-1:]:            # (gpurun merges into gpurun_out/: keep the newest pass only)
        for r in csv.DictReader(open(f)):
            n = label(r["Kernel_Name"])
            if n:
                acc[n][r["Counter_Name"]] += float(r["Counter_Value"]); launches[n].add(r["Dispatch_Id"])
    return acc, {k: len(v) for k, v in launches.items()}


os.makedirs("profiles", exist_ok=True)
st = sorted(glob.glob("gpurun_out/" + TAG + "_stats/*/*kernel_stats.csv"), key=os.path.getmtime)[-1]
shutil.copy(st, "profiles/" + TAG + "_kernel_stats_1Mpairs.csv")
avg_ms = {}
for r in csv.DictReader(open(st)):
    n = label(r["Name"])
    if n:
        avg_ms[n] = float(r["AverageNs"]) / 1e6
f, fl = counters(TAG + "_fetch"); w, wl = counters(TAG + "_write"); q, ql = counters(TAG + "_sq")
rows = []
for k in f:
    fk = f[k]["FETCH_SIZE"] / fl[k]; wk = w[k]["WRITE_SIZE"] / max(1, wl.get(k, 1))
    rows.append((k, fk, wk, (2 * fk + wk) * 1024))
with open("profiles/" + TAG + "_pmc_hbm_1Mpairs.csv", "w") as o:
    o.write("kernel,FETCH_SIZE_KB_per_launch,WRITE_SIZE_KB_per_launch,hbm_bytes_per_launch_(2*FETCH+WRITE)*1024,avg_ms_rocprof\n")
    for r in sorted(rows, key=lambda r: -r[3]):
        o.write("%s,%.3f,%.3f,%.0f,%.3f\n" % (r + (avg_ms.get(r[0], 0.0),)))
cn = ["SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS"]
with open("profiles/" + TAG + "_pmc_sq_1Mpairs.csv", "w") as o:
    o.write("kernel," + ",".join(c + "_per_launch" for c in cn) + ",wait_frac,active_frac\n")
    for k in sorted(q, key=lambda k: -q[k]["SQ_WAVE_CYCLES"]):
        v = [q[k][c] / ql[k] for c in cn]
        o.write(k + "," + ",".join("%.0f" % x for x in v) + ",%.4f,%.4f\n" % (v[2] / max(1.0, v[0]), v[4] / max(1.0, v[0])))
# the first DP class is five kernels launched back to back since round 5 (three band kernels, the jump-free and the general instantiation of the 16-lane template): a
# class-level row = their sum; the DOMINANT kernel is a single kernel (bench.py: roofline.kernel), chosen as the DP kernel that keeps the chip busy longest when it
# runs alone (SQ_BUSY_CYCLES of the single-batch PMC pass) -- the rocprof averages of the default run include the stretch of the side-stream classes beside the next batch
BAND = "k_dp_band<16> + <32> + <64>"
parts = [k for k in ("k_dp_band<16>", "k_dp_band<32>", "k_dp_band<64>") if k in q]
if parts:
    for c in cn:
        q[BAND][c] = sum(q[k][c] / ql[k] for k in parts)
    ql[BAND] = 1
    rp = [r for r in rows if r[0] in parts]
    rows.append((BAND, sum(r[1] for r in rp), sum(r[2] for r in rp), sum(r[3] for r in rp)))
    avg_ms[BAND] = sum(avg_ms.get(k, 0.0) for k in parts)
CLS = "first DP class: " + BAND + " + k_dp<DpTinyJF, 0> + k_dp<DpTiny, 0>"
members = [k for k in (BAND, "k_dp<DpTinyJF, 0>", "k_dp<DpTiny, 0>") if k in q]
for c in cn:
    q[CLS][c] = sum(q[k][c] / ql[k] for k in members)
ql[CLS] = 1
rm = [r for r in rows if r[0] in members]
rows.append((CLS, sum(r[1] for r in rm), sum(r[2] for r in rm), sum(r[3] for r in rm)))
avg_ms[CLS] = sum(avg_ms.get(k, 0.0) for k in members)
single = [k for k in q if (k.startswith("k_dp<") or k == BAND) and k in ("k_dp<DpTinyJF, 0>", "k_dp<DpTiny, 0>", "k_dp<DpMid, 1>", "k_dp<DpSmall, 2>", "k_dp<DpWide, 3>", BAND)]
dom = max(single, key=lambda k: q[k]["SQ_BUSY_CYCLES"] / ql[k])
t = [r for r in rows if r[0] == dom][0]
qs = {c: q[dom][c] / ql[dom] for c in cn}
args = dict(pairs=1048576, levels=5000000, graph="m")
json.dump(dict(args, kernel=dom, kernel_source_hash=bench.kernel_source_hash(), fetch_size_kb=t[1], write_size_kb=t[2], hbm_bytes_per_launch=t[3], rocprof_avg_ms=avg_ms[dom],
               secondary={"wait_frac": qs["SQ_WAIT_ANY"] / qs["SQ_WAVE_CYCLES"], "active_frac": qs["SQ_ACTIVE_INST_ANY"] / qs["SQ_WAVE_CYCLES"],
                          "valu_insts_per_launch": qs["SQ_INSTS_VALU"], "salu_insts_per_launch": qs["SQ_INSTS_SALU"], "lds_insts_per_launch": qs["SQ_INSTS_LDS"],
                          "valu_insts_all_dp_classes_per_launch": sum(q[k]["SQ_INSTS_VALU"] / ql[k] for k in q if (k.startswith("k_dp<") or k.startswith("k_dp_band<")) and k not in (CLS, BAND)),
                          "first_class": {"kernels": CLS, "hbm_bytes_per_launch": [r for r in rows if r[0] == CLS][0][3], "valu_insts_per_launch": q[CLS]["SQ_INSTS_VALU"], "rocprof_avg_ms": avg_ms[CLS]},
                          "source": "profiles/" + TAG + "_pmc_sq_1Mpairs.csv (SQ_WAIT_ANY, SQ_ACTIVE_INST_ANY over SQ_WAVE_CYCLES of the dominant kernel)"},
               note="(2*FETCH_SIZE + WRITE_SIZE)*1024 per launch, separate --pmc passes of `bench.py --steps 1 --warmup 0 --no-extras`; FETCH_SIZE doubled as "
                    "MI355X_MICROARCH.md prescribes for gfx950, WRITE_SIZE uncalibrated"), open("profiles/" + TAG + "_traffic.json", "w"), indent=1)
print("dominant", dom, avg_ms[dom], "ms; hbm bytes/launch %.4g" % t[3])
for r in sorted(rows, key=lambda r: -r[3]):
    print("%-20s %8.2f ms  %.4g bytes  wait %.2f" % (r[0], avg_ms.get(r[0], 0), r[3], q[r[0]]["SQ_WAIT_ANY"] / max(1.0, q[r[0]]["SQ_WAVE_CYCLES"])))
