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


def listing(obj):
    """{function symbol: [(address, mnemonic, operand text, size)]} and the symbols' start addresses, from llvm-objdump -d."""
    out = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", obj], capture_output=True, text=True, check=True).stdout
    funcs, cur = {}, None
    for ln in out.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", ln)
        if m:
            cur = funcs.setdefault(m.group(1), [])
            continue
        m = re.match(r"^\t(\S+)\s*(.*?)\s*//\s*([0-9A-F]+):((?:\s+[0-9A-F]{8})+)\s*(?:<.*>)?\s*$", ln)
        if m and cur is not None:
            cur.append((int(m.group(3), 16), m.group(1), m.group(2), 4 * len(m.group(4).split())))
    return funcs


TERMINATORS = ("s_endpgm", "s_setpc_b64", "s_branch", "s_code_end", "s_trap")


def _base(mn):
    return mn[:-4] if mn.endswith(("_e32", "_e64")) else mn


def _simm(text, bits):
    v = int(text, 0)
    return v - (1 << bits) if v >= 1 << (bits - 1) else v


def padding_of(text, funcs):
    """{function: set of indices into its object instruction list that have NO line in the assembly text}: what the assembler itself put in
    for a .p2align (s_nop) or behind the code (s_code_end).  Same walk as place_function's."""
    lines = text.splitlines()
    pads, i = {}, 0
    while i < len(lines):
        m = re.match(r"^([A-Za-z_.$][\w.$]*):", lines[i])
        if m and m.group(1) in funcs:
            name, ins = m.group(1), funcs[m.group(1)]
            j = i + 1
            while j < len(lines) and not re.match(r"^\.Lfunc_end\d+:", lines[j]):
                j += 1
            k, pad = 0, set()
            for ln in lines[i + 1:j]:
                if not is_instruction(ln):
                    continue
                mn = ln.strip().split()[0]
                while k < len(ins) and ins[k][1] != mn and ins[k][1] in ("s_nop", "s_code_end"):
                    pad.add(k)
                    k += 1
                if k >= len(ins) or ins[k][1] != mn:
                    raise RuntimeError("%s: text and object disagree at %r" % (name, ln.strip()))
                k += 1
            pad.update(range(k, len(ins)))                   # behind the last line of the function: padding up to the next symbol
            pads[name] = pad
            i = j
        else:
            i += 1
    return pads


def check_equivalent(obj_plain, obj_placed, text_plain, text_placed):
    """The placed object must be the plain one plus padding: per function the same instructions in the same order with the same operands,
    where `the same` allows exactly what the pass does -- an _e32 instruction re-encoded as _e64, an `s_nop 0` inserted -- and what follows
    from it -- branch offsets and PC-relative literals that differ in value but name the same instruction.  The ASSEMBLER's padding (the
    nops of a .p2align, the s_code_end behind a function: object instructions that no line of the text stands for) may differ in length.  Anything else -- a changed operand, a missing or reordered
    instruction, a branch that lands elsewhere -- raises.  Returns the number of instructions compared.  This is the property that makes
    the text round-trip of all compiler output safe across a toolchain bump: it is checked on every build."""
    A, B = listing(obj_plain), listing(obj_placed)
    if set(A) != set(B):
        raise RuntimeError("placement changed the set of functions: %s" % sorted(set(A) ^ set(B))[:4])
    pad_a, pad_b = padding_of(text_plain, A), padding_of(text_placed, B)
    # (function, address) -> index of the matched pair at or after that address.  Addresses are section offsets and hipcc gives every
    # function a section of its own, so an address alone names nothing; a branch stays inside its function anyway
    where_a, where_b, pairs_of = {}, {}, {}

    def nop0(ins):
        return ins[1] == "s_nop" and ins[2].strip() == "0"

    for name in A:
        a, b = A[name], B[name]
        ia = ib = 0
        pairs, pend_a, pend_b = [], [], []
        while ia < len(a) or ib < len(b):
            if (ia < len(a) and ib < len(b) and _base(a[ia][1]) == _base(b[ib][1]) and ia not in pad_a.get(name, ()) and ib not in pad_b.get(name, ())):
                for addr in pend_a:
                    where_a[(name, addr)] = len(pairs)
                for addr in pend_b:
                    where_b[(name, addr)] = len(pairs)
                pend_a, pend_b = [], []
                where_a[(name, a[ia][0])] = where_b[(name, b[ib][0])] = len(pairs)
                pairs.append((a[ia], b[ib]))
                ia += 1
                ib += 1
                continue
            if ib < len(b) and (ib in pad_b.get(name, ()) or nop0(b[ib])):      # the assembler's padding, or an `s_nop 0` the pass inserted
                pend_b.append(b[ib][0])
                ib += 1
                continue
            if ia < len(a) and ia in pad_a.get(name, ()):     # the assembler's padding in the plain object (no line of the text stands behind it):
                pend_a.append(a[ia][0])                       # a compiler-emitted nop HAS a line and is never skipped on this side
                ia += 1
                continue
            raise RuntimeError("%s: instruction streams diverge at plain #%d %r / placed #%d %r" % (
                name, ia, a[ia][1:3] if ia < len(a) else None, ib, b[ib][1:3] if ib < len(b) else None))
        for addr in pend_a:
            where_a[(name, addr)] = len(pairs)
        for addr in pend_b:
            where_b[(name, addr)] = len(pairs)
        pairs_of[name] = pairs
    compared = 0
    for name, pairs in pairs_of.items():
        for k, (x, y) in enumerate(pairs):
            compared += 1
            if x[2] == y[2]:
                continue
            mn = _base(x[1])
            try:
                if mn.startswith(("s_cbranch", "s_branch", "s_call_b64")):
                    tx = x[0] + 4 + 4 * _simm(x[2].split(",")[-1].strip(), 16)
                    ty = y[0] + 4 + 4 * _simm(y[2].split(",")[-1].strip(), 16)
                    if x[2].split(",")[:-1] == y[2].split(",")[:-1] and (name, tx) in where_a and where_a[(name, tx)] == where_b.get((name, ty)):
                        continue
                elif mn == "s_add_u32" and k > 0 and pairs[k - 1][0][1] == "s_getpc_b64":
                    ox, oy = [t.strip() for t in x[2].split(",")], [t.strip() for t in y[2].split(",")]
                    tx, ty = x[0] + _simm(ox[-1], 32), y[0] + _simm(oy[-1], 32)
                    # a literal the assembler resolved itself: the target lies in the same section, i.e. (one function per section) in
                    # this function; a reference to another section is a relocation and reads the same in both objects
                    if ox[:-1] == oy[:-1] and (name, tx) in where_a and where_a[(name, tx)] == where_b.get((name, ty)):
                        continue
            except (ValueError, IndexError):
                pass
            raise RuntimeError("%s: instruction #%d differs: plain %r, placed %r" % (name, k, x[1:3], y[1:3]))
    return compared


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
        stats["instructions_compared"] = check_equivalent(obj, obj2, text, placed)
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
