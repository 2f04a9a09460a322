#!/usr/bin/env python3
"""Condenses a tools/profile_all.sh output directory (gpurun_out/<tag>/<workload>/...) into profiles/:

   profiles/<tag>_bench_<w>.json          the bench line of that run
   profiles/<tag>_kernel_stats_<w>.csv    rocprofv3 --kernel-trace --stats summary (verbatim)
   profiles/<tag>_pmc_<w>.json            per kernel: rocprof average duration, launches per step, per-launch PMC means,
                                          HBM-side traffic (FETCH_SIZE doubled per the gfx950 correction + WRITE_SIZE)
   profiles/pmc_traffic.json              HBM bytes per step per workload (read by bench.py for roofline.traffic)

    python tools/summarize_profiles.py gpurun_out/r02p r02
"""
import collections
import csv
import glob
import json
import os
import re
import shutil
import sys

src, tag = sys.argv[1], sys.argv[2]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dst = os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)
# the kernel that runs exactly once per bench step, per workload (launch counts of the others are taken relative to it)
ONCE_PER_STEP = {"cfg2": "ladder_kernel<0, 0, false", "cfg3": "ladder_kernel<1, 1, false",
                 "cfg4": "comb_kernel", "cfg5": "partition_kernel"}


def short(name):
    name = re.sub(r"^void ", "", name)
    name = name.replace("fq::(anonymous namespace)::", "").replace("(anonymous namespace)::", "")
    return name.split("(")[0]


traffic_path = os.path.join(dst, "pmc_traffic.json")
per_workload, build_ids, valu_per_workload = {}, {}, {}
for wdir in sorted(glob.glob(os.path.join(src, "cfg*"))):
    w = os.path.basename(wdir)
    # gpurun merges a call's files INTO gpurun_out/: a tag used twice leaves two runs side by side -- take the newest of each kind
    stats = sorted(glob.glob(os.path.join(wdir, "trace", "*", "*_kernel_stats.csv")), key=os.path.getmtime, reverse=True)
    if not stats:
        continue
    shutil.copy(stats[0], os.path.join(dst, "%s_kernel_stats_%s.csv" % (tag, w)))
    shutil.copy(os.path.join(wdir, "bench.json"), os.path.join(dst, "%s_bench_%s.json" % (tag, w)))
    rows = list(csv.DictReader(open(stats[0])))
    once = [r for r in rows if ONCE_PER_STEP[w] in r["Name"]]
    steps_traced = int(once[0]["Calls"]) if once else None
    kernels = {}
    for r in rows:
        if int(r["Calls"]) < 5 or "fq::" not in r["Name"] and "anonymous" not in r["Name"]:
            continue
        kernels[short(r["Name"])] = {"rocprof_avg_ns": float(r["AverageNs"]), "calls_traced": int(r["Calls"]),
                                     "launches_per_step": round(int(r["Calls"]) / steps_traced, 3) if steps_traced else None,
                                     "share_of_gpu_time_pct": float(r["Percentage"]), "counters": {}}
    for sub in ("pmc_fetch", "pmc_write", "pmc_sq"):
        files = sorted(glob.glob(os.path.join(wdir, sub, "*", "*_counter_collection.csv")), key=os.path.getmtime, reverse=True)
        if not files:
            continue
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        info = {}
        for r in csv.DictReader(open(files[0])):
            k = short(r["Kernel_Name"])
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            info[k] = {f: r[f] for f in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size", "Grid_Size", "Workgroup_Size")}
        for k, ctrs in agg.items():
            if k not in kernels:
                continue
            kernels[k]["dispatch"] = dict(info[k], note="as rocprofv3's counter_collection.csv prints the dispatch packet (VGPR_Count there is NOT the compiler's "
                                                        "register count: fourq_amd/kernel_resources.json has that -- e.g. 256 VGPRs + 40 AGPRs for the headline kernel "
                                                        "where this field says 148)")
            for c, v in ctrs.items():
                kernels[k]["counters"][c] = {"launches": len(v), "mean": sum(v) / len(v), "min": min(v), "max": max(v)}
    step_traffic, step_valu = 0.0, 0.0
    for k, rec in kernels.items():
        c = rec["counters"]
        if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            # rocprofv3 reports both in KiB.  MI355X_MICROARCH.md (HBM section): on gfx950 FETCH_SIZE counts 128-byte requests
            # at 64 bytes for 16-byte-per-lane reads -> double it; WRITE_SIZE is exact.
            fetch_raw, write = c["FETCH_SIZE"]["mean"] * 1024, c["WRITE_SIZE"]["mean"] * 1024
            rec["traffic"] = {"fetch_bytes_raw": fetch_raw, "fetch_bytes_corrected": 2 * fetch_raw, "write_bytes": write,
                              "hbm_bytes_per_launch": 2 * fetch_raw + write}
            step_traffic += (2 * fetch_raw + write) * (rec["launches_per_step"] or 0)
        if "SQ_INSTS_VALU" in c and "SQ_WAVES" in c and c["SQ_WAVES"]["mean"]:
            rec["valu_instructions_per_lane"] = c["SQ_INSTS_VALU"]["mean"] / c["SQ_WAVES"]["mean"]
        if "SQ_INSTS_VALU" in c:
            step_valu += c["SQ_INSTS_VALU"]["mean"] * (rec["launches_per_step"] or 0)          # wave64 instructions per bench step
    bench = json.load(open(os.path.join(wdir, "bench.json")))
    out = {"workload": w, "bench_ms_per_step": bench["ms_per_step"], "bench_kernel_ms": bench["roofline"]["kernel_ms"],
           "rocprof_ms_per_step": sum(r["rocprof_avg_ns"] * (r["launches_per_step"] or 0) for r in kernels.values()) / 1e6,
           "hbm_bytes_per_step": step_traffic, "valu_wave_instructions_per_step": step_valu, "algorithmic_bytes_per_step": bench["roofline"]["algorithmic_bytes_per_launch"],
           "note": "traffic = memory-side (fabric) requests incl. Infinity Cache hits; FETCH_SIZE doubled per the gfx950 correction; "
                   "PMC passes run separately from the kernel trace (10 steps each)", "kernels": kernels}
    with open(os.path.join(dst, "%s_pmc_%s.json" % (tag, w)), "w") as fh:
        json.dump(out, fh, indent=1)
    per_workload[w] = step_traffic
    valu_per_workload[w] = step_valu
    cfg = bench.get("config") or {}
    build_ids[w] = cfg.get("build_id") or (cfg.get("library") or {}).get("build_id")          # the compact line (round 6) / the full record
    print("%s: bench %.4f ms/step, rocprof sum %.4f ms/step, traffic %.1f MB/step (algorithmic %.1f MB)" % (
        w, out["bench_ms_per_step"], out["rocprof_ms_per_step"], step_traffic / 1e6, out["algorithmic_bytes_per_step"] / 1e6))
    for k, r in sorted(kernels.items(), key=lambda kv: -kv[1]["share_of_gpu_time_pct"]):
        print("    %-60s %10.1f us x %.2f/step" % (k[:60], r["rocprof_avg_ns"] / 1e3, r["launches_per_step"] or 0))
if per_workload:
    try:                                     # a partial re-profile (some workloads only) keeps the other workloads' entries
        with open(traffic_path) as fh:
            old = json.load(fh)
        per_workload = dict(old.get("per_workload") or {}, **per_workload)
        valu_per_workload = dict(old.get("valu_wave_instructions_per_step") or {}, **valu_per_workload)
        build_ids = dict(old.get("library_build_ids") or {}, **build_ids)
    except (OSError, ValueError):
        pass
    data = {"hbm_bytes_per_launch": per_workload.get("cfg2"), "per_workload": per_workload, "valu_wave_instructions_per_step": valu_per_workload, "source": "profiles/%s_pmc_<workload>.json" % tag,
            "library_build_ids": build_ids, "library_build_id": build_ids.get("cfg2"),
            "note": "HBM-side bytes per bench step: sum over the step's kernels of (2 x FETCH_SIZE + WRITE_SIZE) per launch x launches per step"}
    with open(traffic_path, "w") as fh:
        json.dump(data, fh, indent=1)
