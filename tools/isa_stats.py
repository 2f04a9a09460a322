#!/usr/bin/env python3
"""Static instruction statistics of the loops of one kernel in a hipcc assembly listing (build box, no GPU).

    hipcc -O3 --offload-arch=gfx950 -std=c++17 --cuda-device-only -S -o /tmp/tu.s fourq_amd/csrc/fourq_chain.hip
    python tools/isa_stats.py /tmp/tu.s 'ladder_kernelILi0ELi2ELb0' [--top 3] [--show]

For every backward branch of the kernel: the instruction count of the loop body (label .. branch, inner loops
included once) and a histogram by mnemonic class.  Used to price a ladder step (DESIGN.md sections 6 and 9): a lone
wave pays ~4 cycles per VALU instruction whatever its kind, so the count is the cost.
"""
import argparse
import collections
import re
import sys


def kernel_body(lines, pattern):
    rx = re.compile(pattern)
    start = None
    for i, ln in enumerate(lines):
        m = re.match(r"^(_Z\w+):", ln)
        if m and rx.search(m.group(1)) and start is None:
            start = i
        elif start is not None and ln.strip().startswith("s_endpgm"):
            return lines[start:i + 1]
    raise SystemExit("kernel matching %r not found" % pattern)


def classify(mn):
    if mn.startswith("v_mad_u64_u32") or mn.startswith("v_mad_i64_i32"):
        return "mad64"
    if mn.startswith(("global_load", "buffer_load", "flat_load", "scratch_load")):
        return "vmem_load"
    if mn.startswith(("global_store", "buffer_store", "flat_store", "scratch_store")):
        return "vmem_store"
    if mn.startswith("ds_"):
        return "lds"
    if mn.startswith("s_nop"):
        return "s_nop"
    if mn.startswith("s_waitcnt"):
        return "s_waitcnt"
    if mn.startswith("s_"):
        return "salu"
    if mn.startswith(("v_accvgpr", "v_mov_b32")):
        return "v_mov/accvgpr"
    if mn.startswith("v_"):
        return mn.split("_e32")[0].split("_e64")[0]
    return "other"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("asm")
    ap.add_argument("kernel", help="regex on the mangled kernel name")
    ap.add_argument("--top", type=int, default=3, help="report the N largest loops")
    ap.add_argument("--show", action="store_true", help="print the body of one loop (--pick, default the largest)")
    ap.add_argument("--pick", type=int, default=0, help="with --show: rank of the loop to print (0 = largest)")
    args = ap.parse_args()
    body = kernel_body(open(args.asm).read().splitlines(), args.kernel)
    labels, instrs = {}, []
    for ln in body:
        t = ln.strip()
        m = re.match(r"^(\.LBB\d+_\d+):", t)
        if m:
            labels[m.group(1)] = len(instrs)
            continue
        if not t or t.startswith((";", ".", "//")) or t.endswith(":"):
            continue
        instrs.append(t.split(";")[0].strip())
    print("kernel: %d instructions" % len(instrs))
    loops = []
    for i, ins in enumerate(instrs):
        m = re.match(r"^s_cbranch_\w+\s+(\.LBB\d+_\d+)", ins) or re.match(r"^s_branch\s+(\.LBB\d+_\d+)", ins)
        if m and m.group(1) in labels and labels[m.group(1)] <= i:
            loops.append((i - labels[m.group(1)] + 1, labels[m.group(1)], i))
    loops.sort(reverse=True)
    for size, lo, hi in loops[:args.top]:
        hist = collections.Counter(classify(x.split()[0]) for x in instrs[lo:hi + 1])
        valu = sum(v for k, v in hist.items() if k.startswith("v_") or k == "mad64")
        print("loop of %d instructions (%d VALU, %d mad64, %d s_nop, %d vmem loads, %d lds):" %
              (size, valu, hist["mad64"], hist["s_nop"], hist["vmem_load"], hist["lds"]))
        print("   " + "  ".join("%s %d" % kv for kv in hist.most_common(24)))
    if args.show and loops:
        _, lo, hi = loops[args.pick]
        print("\n".join(instrs[lo:hi + 1]))


if __name__ == "__main__":
    sys.exit(main())
