#!/usr/bin/env python3
"""Code placement for hipcc's gfx950 assembly: every 8-byte instruction on an 8-byte boundary.

Why (profiles/r04_ladder_step.txt): a VOP3 instruction (v_mad_i64_i32, v_ashrrev_i64, v_lshl_add_u64, ...) is 8 bytes, a VOP1/VOP2/VOPC
_e32 one 4.  An 8-byte instruction that starts at 4 (mod 8) costs a wave measurably more issue time (the same 2 082-instruction ladder step:
3 581 ns aligned, 4 247 ns with its multiply-adds at 4 mod 8), and hipcc places code with no regard for it.  This pass walks the assembly of a
device translation unit and, wherever an 8-byte instruction would start at 4 (mod 8), re-encodes the nearest 4-byte _e32 instruction in
front of it (and behind the previous 8-byte instruction) as its 8-byte _e64 form: same instruction count, same semantics, no padding.  Where
there is no such instruction and a run of at least NOP_RUN 8-byte instructions follows, one s_nop is inserted.

One place must never be touched: s_getpc_b64 and the s_add_u32 / s_addc_u32 behind it, whose literals (sym@rel32@lo+4, @hi+12) are computed
for exactly that spacing -- an s_nop between them sends the call four bytes in front of its target (it did, in an experimental build: illegal
instruction).  Nothing is inserted or re-encoded from an s_getpc_b64 up to the s_addc_u32 that ends the sequence, and the output object is
checked for it (check_pc_relative).

    place_asm.py in.s out.s [--report]

Instruction sizes are not guessed from the text: the input is assembled and disassembled (llvm-objdump prints every encoding), and the k-th
instruction line of a function in the text is the k-th instruction of its section in the object.  The output is assembled again and checked.
"""
import argparse
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
NOP_RUN = 4


def assemble(src, obj):
    cmd = [os.path.join(LLVM, "clang"), "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c", src, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("assembler failed:\n" + r.stderr[-4000:])


def disassemble(obj):
    """{function symbol: [(mnemonic, size)]} from llvm-objdump -d."""
    out = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", obj], capture_output=True, text=True, check=True).stdout
    funcs, cur = {}, None
    for ln in out.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", ln)
        if m:
            cur = funcs.setdefault(m.group(1), [])
            continue
        m = re.match(r"^\t(\S+).*//\s*[0-9A-F]+:((?:\s+[0-9A-F]{8})+)\s*(?:<.*>)?\s*$", ln)
        if m and cur is not None:
            cur.append((m.group(1), 4 * len(m.group(2).split())))
    return funcs


def is_instruction(ln):
    t = ln.strip()
    if not t or t.startswith((";", ".", "//", "#")) or t.endswith(":"):
        return False
    return re.match(r"^[a-z][a-z0-9_]*(\s|$)", t) is not None


def place_function(lines, sizes, name, stats):
    """lines: the text lines of one function (label line excluded); sizes: [(mnemonic, size)] of its instructions in order."""
    # pair every instruction line with its size
    seq, k = [], 0
    for idx, ln in enumerate(lines):
        t = ln.strip()
        m = re.match(r"^\.p2align\s+(\d+)", t)
        if m:
            seq.append(("align", idx, 1 << int(m.group(1))))
            continue
        if not is_instruction(ln):
            continue
        mn = t.split()[0]
        while k < len(sizes) and sizes[k][0] != mn and sizes[k][0] in ("s_nop", "s_code_end"):
            k += 1                                           # padding the assembler put in for a .p2align
        if k >= len(sizes) or sizes[k][0] != mn:
            raise RuntimeError("%s: text and object disagree at %r (object has %r)" % (name, t, sizes[k] if k < len(sizes) else None))
        seq.append(("ins", idx, sizes[k][1], mn))
        k += 1
    # walk, tracking the offset the instruction WILL have
    out = list(lines)
    inserts = {}                                             # line index -> text to insert in front of it
    off, last32, frozen = 0, None, False
    for pos, item in enumerate(seq):
        if item[0] == "align":
            pad = (-off) % item[2]
            off += pad
            last32 = None
            continue
        _, idx, size, mn = item
        if mn == "s_getpc_b64":                              # up to its s_addc_u32: hands off (the literals assume this exact spacing)
            frozen, last32 = True, None
        if frozen:
            if size >= 8 and off % 8 == 4:
                stats["left"] += 1
            if mn in ("s_addc_u32", "s_setpc_b64", "s_swappc_b64"):
                frozen = False
            off += size
            continue
        if size >= 8:
            if off % 8 == 4:
                if last32 is not None:
                    out[last32] = out[last32].replace("_e32", "_e64", 1)
                    off += 4
                    stats["promoted"] += 1
                else:
                    run = 0
                    for nxt in seq[pos:]:
                        if nxt[0] == "ins" and nxt[2] >= 8:
                            run += 1
                        else:
                            break
                    if run >= NOP_RUN:
                        inserts[idx] = "\ts_nop 0\n"
                        off += 4
                        stats["nops"] += 1
                    else:
                        stats["left"] += run
            last32 = None
            stats["wide"] += 1
        else:
            if mn.endswith("_e32"):
                last32 = idx
            elif mn.startswith(("s_cbranch", "s_branch", "s_endpgm", "s_setpc", "s_swappc", "s_barrier")):
                last32 = None                                # do not reach back across control flow
        off += size
    res = []
    for idx, ln in enumerate(out):
        if idx in inserts:
            res.append(inserts[idx])
        res.append(ln)
    return res


def place_text(text, funcs, stats):
    lines = text.splitlines(keepends=True)
    out, i = [], 0
    while i < len(lines):
        m = re.match(r"^([A-Za-z_.$][\w.$]*):", lines[i])
        if m and m.group(1) in funcs:
            name = m.group(1)
            j = i + 1
            while j < len(lines) and not re.match(r"^\.Lfunc_end\d+:", lines[j]):
                j += 1
            out.append(lines[i])
            out += place_function(lines[i + 1:j], funcs[name], name, stats)
            i = j
        else:
            out.append(lines[i])
            i += 1
    return "".join(out)


def misaligned(obj):
    out = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", obj], capture_output=True, text=True, check=True).stdout
    bad = total = 0
    for ln in out.splitlines():
        m = re.match(r"^\t(\S+).*//\s*([0-9A-F]+):((?:\s+[0-9A-F]{8})+)\s*(?:<.*>)?\s*$", ln)
        if m and len(m.group(3).split()) >= 2 and m.group(1) not in ("s_code_end",):
            total += 1
            bad += int(m.group(2), 16) % 8 == 4
    return bad, total


def check_pc_relative(obj):
    """every s_getpc_b64 of the object is followed at once by its s_add_u32 / s_addc_u32 pair (or, for a relaxed branch, by the label-relative
    pair the compiler writes the same way); anything in between breaks the address arithmetic"""
    for name, ins in disassemble(obj).items():
        for k, (mn, _) in enumerate(ins):
            if mn == "s_getpc_b64":
                follow = [m for m, _ in ins[k + 1:k + 3]]
                if follow != ["s_add_u32", "s_addc_u32"]:
                    raise RuntimeError("%s: s_getpc_b64 followed by %s" % (name, follow))


def place_file(src, dst, report=False):
    stats = {"promoted": 0, "nops": 0, "left": 0, "wide": 0}
    with tempfile.TemporaryDirectory() as tmp:
        obj = os.path.join(tmp, "in.o")
        assemble(src, obj)
        check_pc_relative(obj)                               # the INPUT already deviates: do not place what is not understood
        before = misaligned(obj)
        funcs = disassemble(obj)
        with open(src) as fh:
            text = fh.read()
        placed = place_text(text, funcs, stats)
        with open(dst, "w") as fh:
            fh.write(placed)
        obj2 = os.path.join(tmp, "out.o")
        assemble(dst, obj2)
        check_pc_relative(obj2)
        after = misaligned(obj2)
    stats["misaligned_before"], stats["misaligned_after"], stats["wide_total"] = before[0], after[0], after[1]
    if report:
        print("placement: %(wide_total)d 8-byte instructions, %(misaligned_before)d at 4 mod 8 before, %(misaligned_after)d after "
              "(%(promoted)d _e32 -> _e64, %(nops)d s_nop inserted)" % stats, file=sys.stderr)
    return stats


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("src")
    ap.add_argument("dst")
    ap.add_argument("--report", action="store_true")
    a = ap.parse_args()
    place_file(a.src, a.dst, a.report)
