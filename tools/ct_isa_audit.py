#!/usr/bin/env python3
"""Static audit of the constant-time kernels' machine code (build box, no GPU).

    python tools/ct_isa_audit.py            # compiles fourq_ct_fused.hip / fourq_ct_chain.hip to gfx950 assembly and checks them

draft-ladd-cfrg-4q.md:753-758: "Implementations MUST ensure that ... memory addresses accessed do not depend on secret
data" (and no secret-dependent branches).  The secret enters a ladder step as the step's digit and sign, which are
extracted INSIDE the ladder loop (endo_digit / win_window of recode.hip.h: a shift by the loop counter).  So for every loop
of a constant-time kernel that contains multiply-adds (the ladder loops, the comb's column loops) the audit requires:

  1. every memory instruction in the loop (ds_read*, ds_write*, global_/buffer_/scratch_/flat_ load or store) takes its
     address from registers whose backward slice inside the loop consists of address arithmetic (adds, shifts, moves) on
     loop-invariant registers, scalars and constants -- never of a loaded value, a product, a compare or a select -- so the
     address is the same in every iteration up to a fixed stride and cannot carry a digit;
  2. every conditional branch in the loop tests a condition that derives, inside the loop, from scalar registers and constants
     only (hipcc sometimes runs the uniform loop counter through the vector ALU and v_readfirstlane; that is traced and accepted,
     a condition or a v_readfirstlane fed by any other vector register is a finding).

profiles/r03_ct_invariance.txt is the dynamic counterpart (hardware counters identical for every class of scalars).
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "fourq_amd", "csrc")
UNITS = ["fourq_ct_fused.hip", "fourq_ct_chain.hip"]
MEM = ("ds_read", "ds_write", "global_load", "global_store", "buffer_load", "buffer_store", "scratch_load", "scratch_store", "flat_load", "flat_store")


def compile_asm(unit, out):
    hipcc = "/opt/rocm/bin/hipcc"
    cmd = [hipcc, "-O3", "--offload-arch=gfx950", "-std=c++17", "--cuda-device-only", "-S", "-o", out, os.path.join(SRC, unit)]
    return subprocess.Popen(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)


def regs(op):
    """VGPR numbers named by one operand: v12, v[4:7]."""
    m = re.fullmatch(r"v(\d+)", op)
    if m:
        return {int(m.group(1))}
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", op)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    return set()


def kernels(lines):
    name, body = None, []
    for ln in lines:
        m = re.match(r"^(_Z\w+):", ln)
        if m:
            name, body = m.group(1), []
            continue
        if name is not None:
            body.append(ln)
            if ln.strip().startswith("s_endpgm"):
                yield name, body
                name = None


def parse(body):
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
    return labels, instrs


def operands(ins):
    parts = ins.split(None, 1)
    return [o.strip() for o in parts[1].split(",")] if len(parts) > 1 else []


ADDRESS_OPS = ("v_add_u32", "v_add_co_u32", "v_addc_co_u32", "v_add_nc_u32", "v_add3_u32", "v_lshl_add_u32", "v_lshl_add_u64", "v_add_lshl_u32",
               "v_mov_b32", "v_mov_b64", "v_or_b32", "v_lshlrev_b32", "v_lshlrev_b64", "v_accvgpr_read", "v_accvgpr_write", "v_mad_u32_u24", "v_mul_u32_u24",
               "v_and_b32", "v_or3_b32", "v_lshl_or_b32", "v_and_or_b32")
LADDER_MADS = (1300, 1200, 1800, 650)      # static multiply-adds of one ladder step (DBL + ADD; 4 DBL in an inner loop + ADD; since round 4's asm bodies:
                                           # 3 DBL in an inner loop + DBL-with-T + ADD = 500 + 600 + 700) / one comb column / a pair-lane step
# four lanes per element (pair_kernel<..., 4>): a ladder step is known by its multiply-adds AND its exchanges between the element's two pairs
# (v_mov_b32_dpp quad_perm:[2,3,0,1], five per shared result): MUL_endo 350 / 35 (7 results), MUL_windowed 500 / 50 (the loop of three doublings
# counted once: 3 + 4 + 3 results).  The table-building loops of the same kernels share 3 results per addition (300 / 15).
QUAD_STEP = ((350, 35), (500, 50))


def defs_of(ins):
    """VGPRs (and AGPRs, numbered from 1000) an instruction writes."""
    mn, ops = ins.split()[0], operands(ins)
    if not ops or mn.startswith(("s_", "ds_write", "global_store", "buffer_store", "scratch_store", "flat_store", "v_cmp", "v_cmpx")):
        return set()
    out = set(regs(ops[0])) | {1000 + r for r in regs(ops[0].replace("a", "v", 1))} if ops[0].startswith("a") else set(regs(ops[0]))
    if mn.startswith(("v_mad_u64_u32", "v_mad_i64_i32")) and len(ops) > 1:
        out |= regs(ops[1])
    return out


def all_regs(op):
    if op.startswith("a"):
        return {1000 + r for r in regs(op.replace("a", "v", 1))}
    return regs(op)


def slice_is_invariant(seg, k, reg, depth=0, seen=None):
    """True when register `reg` as read by instruction k of the loop body `seg` is computed only from loop-invariant registers,
    scalars and constants through address arithmetic -- so that it cannot carry the step's digit (which is extracted in the
    loop from vector registers by shifts that depend on the loop counter).  Returns (ok, reason)."""
    seen = seen if seen is not None else set()
    if (k, reg) in seen or depth > 40:
        return True, ""
    seen.add((k, reg))
    n = len(seg)
    for back in range(1, n + 1):                       # the most recent definition, wrapping around the back edge once
        j = (k - back) % n
        if reg in defs_of(seg[j]):
            ins = seg[j]
            mn, ops = ins.split()[0], operands(ins)
            base = mn.split("_e32")[0].split("_e64")[0].split("_dpp")[0]
            if not base.startswith(ADDRESS_OPS):
                return False, "v%d <- `%s` (not address arithmetic)" % (reg, ins)
            for src in ops[1:]:
                for r in all_regs(src):
                    ok, why = slice_is_invariant(seg, j, r, depth + 1, seen)
                    if not ok:
                        return False, why
            return True, ""
    return True, ""                                     # never written in the loop: loop-invariant


def names(op):
    """register names an operand mentions: VGPR numbers (ints), 's12', 'vcc', 'exec'"""
    op = op.strip()
    if regs(op):
        return set(regs(op))
    m = re.fullmatch(r"s(\d+)", op)
    if m:
        return {"s" + m.group(1)}
    m = re.fullmatch(r"s\[(\d+):(\d+)\]", op)
    if m:
        return {"s%d" % i for i in range(int(m.group(1)), int(m.group(2)) + 1)}
    if op.startswith("vcc"):
        return {"vcc"}
    if op.startswith("exec"):
        return {"exec"}
    return set()


def dests_and_sources(ins):
    mn, ops = ins.split()[0], operands(ins)
    if not ops or mn.startswith(("s_cbranch", "s_branch", "s_waitcnt", "s_nop", "ds_write", "global_store", "buffer_store", "scratch_store", "flat_store")):
        return set(), set()
    ndst = 1
    if re.search(r"_co_u32|_co_ci_u32|v_addc|v_subb|v_mad_u64_u32|v_mad_i64_i32|v_div_scale", mn) and len(ops) > 2:
        ndst = 2
    if mn.startswith(("s_cmp", "s_bitcmp")):
        return {"scc"}, set().union(*[names(o) for o in ops])
    d = set().union(*[names(o) for o in ops[:ndst]])
    srcs = set().union(*[names(o) for o in ops[ndst:]]) if len(ops) > ndst else set()
    if mn.startswith("v_cndmask_b32_e32") or "vcc" in ins.split(None, 1)[1] and mn.startswith(("v_addc", "v_subb")):
        srcs |= {"vcc"}
    return d, srcs


def is_public(seg, k, name, depth=0, seen=None):
    """True when the value of `name` read by instruction k derives, inside the loop, from scalar registers and constants only
    (a uniform, public quantity such as the loop counter).  A vector register that the loop does not write may hold anything
    (the digit planes live in such registers), so it is NOT public."""
    seen = seen if seen is not None else set()
    if (k, name) in seen or depth > 30:
        return True
    seen.add((k, name))
    n = len(seg)
    for back in range(1, n + 1):
        j = (k - back) % n
        d, srcs = dests_and_sources(seg[j])
        if name in d:
            return all(is_public(seg, j, s_, depth + 1, seen) for s_ in srcs)
    return not isinstance(name, int)            # loop-invariant: scalars are kernel arguments / uniform values; vector registers are not trusted


def audit_kernel(name, body):
    labels, instrs = parse(body)
    loops = []
    for i, ins in enumerate(instrs):
        m = re.match(r"^s_c?branch\w*\s+(\.LBB\d+_\d+)", ins)
        if m and m.group(1) in labels and labels[m.group(1)] <= i:
            loops.append((labels[m.group(1)], i))

    def mads_of(lo, hi):
        return sum(1 for x in instrs[lo:hi + 1] if x.startswith(("v_mad_u64_u32", "v_mad_i64_i32")))
    # pair_kernel<ALGO, DH, CT, FIXED, 4, MIXED>; comb_quad_kernel<CT, 4>: the comb's quad step = doubling + mixed addition, the same 350 / 35
    # (comb_quad_kernel<CT, 2> is the two-lane code: 650 multiply-adds, as a pair-lane ladder step)
    if ("pair_kernel" in name and "ELi4ELb" in name) or ("comb_quad_kernel" in name and "ELi4EEE" in name):
        def shares(lo, hi):
            return sum(1 for x in instrs[lo:hi + 1] if "quad_perm:[2,3,0,1]" in x)
        ladder = [(lo, hi) for lo, hi in loops if (mads_of(lo, hi), shares(lo, hi)) in QUAD_STEP]
    elif "comb_quad_kernel" in name:        # <CT, 2>: doubling (250) + T and the addition's seven products (350); hipcc also latches the addition alone (350): the whole column is the subject
        ladder = [(lo, hi) for lo, hi in loops if mads_of(lo, hi) == 600]
    else:
        ladder = [(lo, hi) for lo, hi in loops if mads_of(lo, hi) in LADDER_MADS]
    # the ladder-step / comb-column loops are the INNERMOST loops with a whole step's multiply-adds (the element loop around
    # them, which for a bare MUL kernel has the same count, is public control flow: it indexes by element number)
    ladder = [(lo, hi) for lo, hi in ladder if not any((l2, h2) != (lo, hi) and lo <= l2 and h2 <= hi for l2, h2 in ladder)]
    problems = []
    for lo, hi in ladder:
        seg = instrs[lo:hi + 1]
        for k, ins in enumerate(seg):
            mn = ins.split()[0]
            ops = operands(ins)
            if mn.startswith(("v_readlane", "v_readfirstlane")):
                if not all(is_public(seg, k, r) for r in regs(ops[1])):
                    problems.append("%s: loop @%d: `%s` moves lane data that is not a loop counter into a scalar" % (name, lo, ins))
            if mn.startswith(MEM):
                addr_ops = [ops[0]] if ("store" in mn or mn.startswith("ds_write")) else ops[1:2]
                for a_op in addr_ops:
                    for r in regs(a_op):
                        ok, why = slice_is_invariant(seg, k, r)
                        if not ok:
                            problems.append("%s: loop @%d: address of `%s`: %s" % (name, lo, ins, why))
            m = re.match(r"^s_cbranch_(vcc|exec|scc)", mn)
            if m and not is_public(seg, k, m.group(1)):
                problems.append("%s: loop @%d: `%s`: the branch condition derives from vector data" % (name, lo, ins))
    return len(ladder), problems


def audit_units(units, tmp=None):
    """Compile `units` to assembly and audit every ladder / comb / queue kernel.  Returns (kernels, loops, findings)."""
    tmp = tmp or os.environ.get("TMPDIR", "/tmp")
    procs = []
    for u in units:
        out = os.path.join(tmp, "ct_audit_" + u.replace(".hip", ".s"))
        procs.append((u, out, compile_asm(u, out)))
    rows, problems = [], []
    for u, out, p in procs:
        if p.wait() != 0:
            raise SystemExit("hipcc failed on %s" % u)
        for name, body in kernels(open(out).read().splitlines()):
            if not any(k in name for k in ("ladder_kernel", "comb_kernel", "comb_quad_kernel", "mixed_queue_kernel", "mixed_ct_tail_kernel", "pair_kernel")):
                continue
            checked, bad = audit_kernel(name, body)
            rows.append((u, name, checked, len(bad)))
            problems += bad
    return rows, problems


def main():
    control = "--control" in sys.argv           # the default-mode kernels, where a digit IS an address: the audit must object
    rows, problems = audit_units(["fourq_amd.hip", "fourq_chain.hip"] if control else UNITS)
    for u, name, checked, bad in rows:
        print("%-20s %-64s %d ladder loop(s), %d finding(s)" % (u, name[20:84], checked, bad))
    for p in problems[:40]:
        print("FINDING:", p[:400])
    loops = sum(r[2] for r in rows)
    print("%s kernels audited: %d, ladder loops: %d, findings: %d" % ("default-mode (control)" if control else "constant-time", len(rows), loops, len(problems)))
    if control:
        return 0 if problems else 1
    return 1 if problems or loops == 0 else 0


if __name__ == "__main__":
    sys.exit(main())
