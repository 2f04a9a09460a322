#!/usr/bin/env python3
"""Condenses tools/ct_invariance.sh's rocprofv3 output: per selection mode and kernel, each counter's mean per launch for every class of
secret scalars and the spread between classes.  Only the launches after the probe's "subject" marker count: the last reps x 5
kernels of each run (the probe's set-up runs in the default mode)."""
import collections
import csv
import glob
import os
import sys

root = sys.argv[1]
CLASSES = ["random", "zero", "ones", "same"]
SUBJECT = ("ladder_kernel", "comb_kernel", "comb_quad_kernel", "normalize_kernel", "prep_kernel", "pair_kernel")


def is_ct_kernel(kname):
    """The CT template argument: the last one of ladder_kernel<ALGO, SRC, DH, DEFER, CT> and comb_kernel<.., CT>, the third of
    pair_kernel<ALGO, DH, CT, FIXED, LPE, MIXED>, the first of comb_quad_kernel<CT, LPE>; normalize_kernel has no table."""
    if kname.startswith("normalize_kernel"):
        return True
    args = [a.strip() for a in kname[kname.find("<") + 1:kname.rfind(">")].split(",")]
    if kname.startswith("pair_kernel"):
        return len(args) > 2 and args[2] == "true"
    if kname.startswith("comb_quad_kernel"):          # comb_quad_kernel<CT, lanes per element>
        return bool(args) and args[0] == "true"
    return bool(args) and args[-1] == "true"


def short(name):
    n = name.replace("void fq::(anonymous namespace)::", "").replace("fq::(anonymous namespace)::", "").replace("(anonymous namespace)::", "")
    return n.split("(")[0]


def rows(pattern):
    files = glob.glob(pattern)
    return list(csv.DictReader(open(files[0]))) if files else []


def tail_launches(rs, key, per_kernel):
    """keep, per kernel name, the LAST `per_kernel[name]` launches: those are the probe's measured launches"""
    by = collections.defaultdict(list)
    for r in rs:
        by[short(r[key])].append(r)
    return by


for mode in ("ct", "default"):
    if not glob.glob(os.path.join(root, mode + "_*_trace")):
        continue
    print("## selection mode: %s" % ("constant-time (fourq_ctx_set_ct_select = 1)" if mode == "ct" else "default (digit = table address, as the reference)"))
    table = collections.defaultdict(dict)          # (kernel, counter) -> {class: mean}
    for cls in CLASSES:
        trace = rows(os.path.join(root, "%s_%s_trace" % (mode, cls), "*", "*_kernel_trace.csv"))
        by = tail_launches(trace, "Kernel_Name", None)
        reps = 4
        for kname, rs in by.items():
            if not kname.startswith(SUBJECT):
                continue
            d = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rs][-reps:]
            table[(kname, "duration_us")][cls] = sum(d) / len(d) / 1e3
        for tag in ("fetch", "sq"):
            cc = rows(os.path.join(root, "%s_%s_%s" % (mode, cls, tag), "*", "*_counter_collection.csv"))
            agg = collections.defaultdict(list)
            for r in cc:
                kname = short(r["Kernel_Name"])
                if kname.startswith(SUBJECT):
                    agg[(kname, r["Counter_Name"])].append(float(r["Counter_Value"]))
            for key, vals in agg.items():
                vals = vals[-reps:]
                table[key][cls] = sum(vals) / len(vals)
    print("%-44s %-22s %14s %14s %14s %14s   %s" % ("kernel", "counter (mean per launch)", *CLASSES, "max spread"))
    for (kname, ctr), per in sorted(table.items()):
        if mode == "ct" and not is_ct_kernel(kname):
            continue                                   # the probe's set-up launches (default-mode kernels), not the subject
        vals = [per.get(c) for c in CLASSES]
        have = [v for v in vals if v is not None]
        if len(have) < 2:
            continue
        lo, hi = min(have), max(have)
        spread = (hi - lo) / hi * 100 if hi else 0.0
        print("%-44s %-22s %s   %6.2f %%" % (kname[:44], ctr, " ".join("%14.1f" % v if v is not None else "%14s" % "-" for v in vals), spread))
    print()
