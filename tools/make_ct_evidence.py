#!/usr/bin/env python3
"""Assembles profiles/r03_ct_invariance.txt and profiles/r03_kernel_stats_small_batches.csv from the outputs of tools/runs/r03_ct_evidence.sh
(gpurun_out/r03c) and tools/runs/r03_small_stats.sh (gpurun_out/r03s); the explanatory header of the evidence file is kept.  Build box.
    python tools/make_ct_evidence.py <build id>"""
import csv, glob, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.chdir(ROOT)
BUILD = sys.argv[1]
TAG = sys.argv[2] if len(sys.argv) > 2 else "r03"          # round tag: reads gpurun_out/<tag>c[_ct]/..., writes profiles/<tag>_ct_invariance.txt
SRC = "gpurun_out/r03c" if TAG == "r03" else "gpurun_out/%sc_ct" % TAG
pct = lambda l: float(re.search(r"([\d.]+) %\s*$", l).group(1))
old = open("profiles/r03_ct_invariance.txt").read()
head = old[:old.index("## selection mode: constant-time")]
head = re.sub(r"build [0-9a-f]{16}: selection by a binary tree of masked selects\):", "build %s: selection by a binary tree of masked selects):" % BUILD, head)
full = open(SRC + "/full.txt").read()
full = full[full.index("## selection mode: constant-time"):]
rev = [l for l in open(SRC + "/reversed.txt").read().splitlines() if "duration_us" in l]
small = lambda path: "\n".join(l for l in open(path).read().splitlines() if l.startswith(("pair_kernel", "comb_quad_kernel")))
ct_part = full[:full.index("## selection mode: default")]
fs = [pct(l) for l in ct_part.splitlines() if "FETCH_SIZE" in l or "WRITE_SIZE" in l]
others = [pct(l) for l in ct_part.splitlines() if re.search(r"%\s*$", l) and not any(k in l for k in ("duration", "FETCH_SIZE", "WRITE_SIZE"))]
du = [pct(l) for l in full.splitlines() if "duration_us" in l]
assert max(others) == 0.0, "an instruction / LDS counter differs between scalar classes in constant-time mode"
head = re.sub(r"up to [\d.]+ % on FETCH_SIZE /", "up to %.1f %% on FETCH_SIZE /" % max(fs), head)
head = re.sub(r"the random class runs [\d-]+ % slower \(either mode\)", "the random class runs %d-%d %% slower (either mode)" % (round(min(du)), round(max(du))), head)
q, p2 = small(SRC + "/quad.txt"), small(SRC + "/pair.txt")
both = (q + "\n" + p2).splitlines()
counters = [l for l in both if "duration" not in l and "FETCH_SIZE" not in l]
assert all(l.rstrip().endswith("0.00 %") for l in counters), [l for l in counters if not l.rstrip().endswith("0.00 %")][:3]
fetch = [pct(l) for l in both if "FETCH_SIZE" in l]
dd = [pct(l) for l in both if "duration" in l]
kernels = sorted({re.match(r"((?:pair|comb_quad)_kernel<[^>]*>)", l).group(1) for l in both})
new = head + full.rstrip("\n") + "\n\n## durations only, classes run in the order same, ones, zero, random (columns as above): the random class stays the slow one\n" + "\n".join(rev) + "\n\n"
new += """## small batches, constant-time mode, same four classes of scalars (columns random / zero / ones / same)
## (pair_kernel<ALGO, DH, CT = true, FIXED, lanes per element, MIXED>: variable-base MUL_endo, fixed-base MUL_endo and MUL_windowed, DH_endo; comb_quad_kernel<true, lanes>: key generation)
## four lanes per element (batches of at most a quarter generation: here 4 096 elements)
""" + q + """
## two lanes per element (batches between a quarter and half a generation: here 20 000 elements)
""" + p2 + "\n"
new += "# small batches: every instruction counter (VALU, LDS, VMEM), LDS busy and bank-conflict cycles and WRITE_SIZE identical over the four classes in all %d\n# kernels; FETCH_SIZE within %.2f %%; durations within %.1f-%.1f %%.\n" % (len(kernels), max(fetch), min(dd), max(dd))
if TAG != "r03":
    new = new.replace("# r03: counter evidence", "# %s: counter evidence" % TAG, 1).replace("tools/runs/r03_ct_evidence.sh", "tools/runs/%s_ct_evidence.sh" % TAG)
    new = new.replace("selection by a binary tree of masked selects):", "selection by a binary tree of masked selects; since round 4 the ladders' doubling and addition\n# are the generated asm bodies of ladder_asm.hip.h, fed by the scan's select trees):", 1)
open("profiles/%s_ct_invariance.txt" % TAG, "w").write(new)
if TAG != "r03":
    print("wrote profiles/%s_ct_invariance.txt (%d constant-time small-batch kernels; FETCH/WRITE spread <= %.2f %% full size, %.2f %% small; durations %.1f-%.1f %% / %.1f-%.1f %%)" % (
        TAG, len(kernels), max(fs), max(fetch), min(du), max(du), min(dd), max(dd)))
    sys.exit(0)
out = ["# r03: rocprofv3 --kernel-trace --stats of the small-batch kernels (tools/runs/r03_small_stats.sh: tools/ct_probe.py launches variable-base MUL_endo, fixed-base MUL_endo and",
       "# MUL_windowed, comb keygen and DH_endo 20 times each on n elements; one MI355X, build %s).  pair_kernel<ALGO (0 endo, 1 windowed), DH, CT, FIXED, lanes per element, MIXED>." % BUILD,
       "# Average duration per launch agrees with tools/quad_probe.py's HIP-event figures (profiles/r03_quadlane.txt).  Single calls in the list are the probe's set-up launches.",
       "mode,n,kernel,calls,average_us,min_us,max_us"]
for mode in ("default", "ct"):
    for n in (4096, 20000):
        f = glob.glob("gpurun_out/r03s/%s_%d/*/*_kernel_stats.csv" % (mode, n))[0]
        for r in csv.DictReader(open(f)):
            m = re.search(r"((?:pair|comb|comb_quad)_kernel<[^>]*>)", r["Name"])
            if m:
                out.append("%s,%d,\"%s\",%s,%.1f,%.1f,%.1f" % ("constant-time" if mode == "ct" else "default", n, m.group(1), r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
open("profiles/r03_kernel_stats_small_batches.csv", "w").write("\n".join(out) + "\n")
print("constant-time kernels in the small-batch tables: %d; FETCH/WRITE spread <= %.2f %% (full size), %.2f %% (small); durations %.1f-%.1f %% / %.1f-%.1f %%" % (len(kernels), max(fs), max(fetch), min(du), max(du), min(dd), max(dd)))
