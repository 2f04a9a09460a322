#!/usr/bin/env python3
"""Condenses a tools/profile_bench.sh output directory (gpurun_out/<tag>) into profiles/:
   profiles/<tag>_kernel_stats.csv   rocprofv3 --kernel-trace --stats summary (verbatim)
   profiles/<tag>_pmc.json           per-launch means of the PMC counters of the dominant kernel
   profiles/pmc_traffic.json         HBM bytes per launch (read by bench.py for roofline.traffic)

    python tools/summarize_profile.py gpurun_out/r01a r01
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

src, tag = sys.argv[1], sys.argv[2]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dst = os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)
KERNEL = sys.argv[3] if len(sys.argv) > 3 else "ladder_kernel<0, 0, false, false>"

stats = glob.glob(os.path.join(src, "trace", "*", "*_kernel_stats.csv"))[0]
shutil.copy(stats, os.path.join(dst, "%s_kernel_stats.csv" % tag))
shutil.copy(os.path.join(src, "bench.json"), os.path.join(dst, "%s_bench.json" % tag))
avg_ns = None
for row in csv.DictReader(open(stats)):
    if KERNEL in row["Name"]:
        avg_ns = float(row["AverageNs"])

pmc = {"kernel": KERNEL, "rocprof_avg_ns": avg_ns, "counters": {}}
info = {}
for sub in ("pmc_fetch", "pmc_write", "pmc_sq"):
    files = glob.glob(os.path.join(src, sub, "*", "*_counter_collection.csv"))
    if not files:
        continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(files[0])):
        if KERNEL in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
            info = {k: r[k] for k in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size", "Grid_Size", "Workgroup_Size")}
    for k, v in agg.items():
        pmc["counters"][k] = {"launches": len(v), "mean": sum(v) / len(v), "min": min(v), "max": max(v)}
pmc["dispatch"] = info
c = pmc["counters"]
if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
    # rocprofv3 reports both in KiB.  MI355X_MICROARCH.md (HBM section): on gfx950 FETCH_SIZE counts 128-byte
    # requests at 64 bytes for 16-byte-per-lane reads -> double it; WRITE_SIZE is exact.
    fetch_raw = c["FETCH_SIZE"]["mean"] * 1024
    write = c["WRITE_SIZE"]["mean"] * 1024
    traffic = {"fetch_bytes_raw": fetch_raw, "fetch_bytes_corrected": 2 * fetch_raw, "write_bytes": write,
               "hbm_bytes_per_launch": 2 * fetch_raw + write,
               "note": "memory-side (fabric) requests incl. Infinity Cache hits; FETCH_SIZE doubled per the gfx950 correction"}
    pmc["traffic"] = traffic
    with open(os.path.join(dst, "pmc_traffic.json"), "w") as fh:
        json.dump(dict(traffic, source="profiles/%s_pmc.json" % tag), fh, indent=1)
with open(os.path.join(dst, "%s_pmc.json" % tag), "w") as fh:
    json.dump(pmc, fh, indent=1)
print(json.dumps(pmc, indent=1))
