#!/usr/bin/env python3
"""Reads a rocprofv3 --kernel-trace --memory-copy-trace output directory of ONE host-array call pattern (tools/pipeline_probe.py) and
prints, for the last call in the trace, where every chunk's copies and kernels sit: start / end of each H2D copy, kernel and D2H copy
relative to the first H2D byte, the idle gaps of the kernel stream between chunks, and the sums that say what bounds the call (kernel
busy time, copy busy time per direction, bubbles).

    python tools/pipeline_timeline.py <dir with *_kernel_trace.csv and *_memory_copy_trace.csv> [--calls N] [--ladder SUBSTRING]
"""
import csv
import glob
import os
import sys


def rows(pattern, root):
    out = []
    for path in glob.glob(os.path.join(root, "**", pattern), recursive=True):
        with open(path) as fh:
            out += list(csv.DictReader(fh))
    return out


def main(argv):
    root = argv[0]
    calls = int(argv[argv.index("--calls") + 1]) if "--calls" in argv else 4
    ladder = argv[argv.index("--ladder") + 1] if "--ladder" in argv else "ladder_kernel"
    ks = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows("*kernel_trace.csv", root)]
    cs = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Direction"]) for r in rows("*memory_copy_trace.csv", root)]
    ks.sort()
    cs.sort()
    lad = [k for k in ks if ladder in k[2]]
    if not lad:
        print("no kernel matching", ladder)
        return 1
    # a call = a burst of ladder kernels: split the ladder launches where the gap between them is > 1.5 ms
    bursts, cur = [], [lad[0]]
    for k in lad[1:]:
        if k[0] - cur[-1][1] > 1_500_000:
            bursts.append(cur)
            cur = []
        cur.append(k)
    bursts.append(cur)
    per = max(len(b) for b in bursts)
    full = [b for b in bursts if len(b) == per]
    print("%d bursts of ladder launches, %d with %d launches each; showing the last %d" % (len(bursts), len(full), per, min(calls, len(full))))
    for b in full[-calls:]:
        t_lo, t_hi = b[0][0] - 3_000_000, b[-1][1] + 3_000_000
        h2d = [c for c in cs if t_lo < c[0] < t_hi and "HOST_TO_DEVICE" in c[2].upper().replace("MEMORY_COPY_", "")]
        d2h = [c for c in cs if t_lo < c[0] < t_hi and "DEVICE_TO_HOST" in c[2].upper().replace("MEMORY_COPY_", "")]
        # keep only copies adjacent to this burst (first H2D no earlier than 2 ms before the first ladder, last D2H no later than 2 ms after)
        h2d = [c for c in h2d if c[0] > b[0][0] - 2_000_000 and c[0] < b[-1][1]]
        d2h = [c for c in d2h if c[1] < b[-1][1] + 2_000_000 and c[0] > b[0][0]]
        allk = [k for k in ks if b[0][0] - 500_000 < k[0] < b[-1][1] + 500_000 and "copyBuffer" not in k[2]]
        t0 = min([c[0] for c in h2d] + [allk[0][0]])
        t1 = max([c[1] for c in d2h] + [allk[-1][1]])
        us = lambda t: (t - t0) / 1e3
        busy = sum(k[1] - k[0] for k in allk)
        gaps = [allk[i + 1][0] - allk[i][1] for i in range(len(allk) - 1)]
        print("call: first H2D byte -> last D2H byte %.3f ms | kernels: first start %.0f us, last end %.0f us, busy %.3f ms, %d launches, "
              "idle between them %.3f ms (max gap %.0f us) | after the last kernel %.0f us" % (
                  (t1 - t0) / 1e6, us(allk[0][0]), us(allk[-1][1]), busy / 1e6, len(allk), sum(gaps) / 1e6, max(gaps) / 1e3 if gaps else 0, (t1 - allk[-1][1]) / 1e3))
        print("      H2D: %d copies, busy %.3f ms, last end %.0f us | D2H: %d copies, busy %.3f ms, first start %.0f us" % (
            len(h2d), sum(c[1] - c[0] for c in h2d) / 1e6, us(max(c[1] for c in h2d)) if h2d else 0,
            len(d2h), sum(c[1] - c[0] for c in d2h) / 1e6, us(min(c[0] for c in d2h)) if d2h else 0))
        if "--detail" in argv:
            ev = [(k[0], k[1], "K " + k[2][:60]) for k in allk] + [(c[0], c[1], "H2D") for c in h2d] + [(c[0], c[1], "D2H") for c in d2h]
            for a, e, name in sorted(ev):
                print("        %9.1f .. %9.1f us  (%7.1f)  %s" % (us(a), us(e), (e - a) / 1e3, name))
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
