#!/usr/bin/env python3
"""An interpreter for the instruction subset the generated bodies use (tools/asmgen/gen_ladder_step.py): runs a body's text on Python
integers, lane by lane, so that the CPU-only suite can check every generated body against the oracle's formulas WITHOUT a GPU
(tests/test_asm_bodies.py).  Semantics per the gfx950 ISA for the forms the generator emits; anything else raises.

    regs = run(lines, {"%0": limb, ...})        # registers are 32-bit values; "%N" operands, "vN" temporaries, "v[a:b]" pairs
"""
import re

M32, M64 = (1 << 32) - 1, (1 << 64) - 1


def _s32(x):
    return x - (1 << 32) if x & (1 << 31) else x


def _s64(x):
    return x - (1 << 64) if x & (1 << 63) else x


class Machine:
    def __init__(self, regs):
        self.r = dict(regs)

    def get(self, op):
        op = op.strip()
        if op in self.r:
            return self.r[op]
        if re.fullmatch(r"-?\d+", op):
            return int(op) & M32
        if re.fullmatch(r"0x[0-9a-fA-F]+", op):
            return int(op, 16) & M32
        if re.fullmatch(r"v\d+|%\d+", op):
            raise KeyError("read of %s before it was written" % op)
        raise ValueError("operand %r" % op)

    def get64(self, op):
        op = op.strip()
        if op == "0":
            return 0
        m = re.fullmatch(r"v\[(\d+):(\d+)\]", op)
        if not m or int(m.group(2)) != int(m.group(1)) + 1 or int(m.group(1)) % 2:
            raise ValueError("64-bit operand %r" % op)
        return self.get("v" + m.group(1)) | (self.get("v" + m.group(2)) << 32)

    def put(self, op, value):
        self.r[op.strip()] = value & M32

    def put64(self, op, value):
        m = re.fullmatch(r"v\[(\d+):(\d+)\]", op.strip())
        self.r["v" + m.group(1)] = value & M32
        self.r["v" + m.group(2)] = (value >> 32) & M32


def run(lines, regs):
    m = Machine(regs)
    for ln in lines:
        ln = ln.strip()
        if not ln or ln.startswith("."):
            continue
        mn, rest = ln.split(None, 1)
        extra = None
        if " bitop3:" in rest:
            rest, extra = rest.split(" bitop3:")
        ops = [o.strip() for o in rest.split(",")]
        base = re.sub(r"_e(32|64)$", "", mn)
        if base == "v_mov_b32":
            m.put(ops[0], m.get(ops[1]))
        elif base == "v_add_u32":
            m.put(ops[0], m.get(ops[1]) + m.get(ops[2]))
        elif base == "v_sub_u32":
            m.put(ops[0], m.get(ops[1]) - m.get(ops[2]))
        elif base == "v_lshlrev_b32":
            m.put(ops[0], m.get(ops[2]) << (m.get(ops[1]) & 31))
        elif base == "v_lshrrev_b32":
            m.put(ops[0], m.get(ops[2]) >> (m.get(ops[1]) & 31))
        elif base == "v_and_b32":
            m.put(ops[0], m.get(ops[1]) & m.get(ops[2]))
        elif base == "v_xor_b32":
            m.put(ops[0], m.get(ops[1]) ^ m.get(ops[2]))
        elif base == "v_lshl_add_u32":
            m.put(ops[0], (m.get(ops[1]) << (m.get(ops[2]) & 31)) + m.get(ops[3]))
        elif base == "v_bitop3_b32":
            if int(extra, 16) != 0xCA:
                raise ValueError("bitop3 table " + extra)
            a, b, c = m.get(ops[1]), m.get(ops[2]), m.get(ops[3])
            m.put(ops[0], (a & b) | (~a & c))
        elif base == "v_alignbit_b32":
            m.put(ops[0], ((m.get(ops[1]) << 32 | m.get(ops[2])) >> (m.get(ops[3]) & 31)))
        elif base in ("v_mad_i64_i32", "v_mad_u64_u32"):
            if ops[1] != "vcc":
                raise ValueError("carry-out operand " + ops[1])
            a, b, c = m.get(ops[2]), m.get(ops[3]), m.get64(ops[4])
            if base == "v_mad_i64_i32":
                a, b, c = _s32(a), _s32(b), _s64(c)
            m.put64(ops[0], (a * b + c) & M64)
        elif base == "v_ashrrev_i64":
            m.put64(ops[0], (_s64(m.get64(ops[2])) >> (m.get(ops[1]) & 63)) & M64)
        elif base == "v_lshrrev_b64":
            m.put64(ops[0], m.get64(ops[2]) >> (m.get(ops[1]) & 63))
        elif base == "v_lshl_add_u64":
            m.put64(ops[0], ((m.get64(ops[1]) << (m.get(ops[2]) & 63)) + m.get64(ops[3])) & M64)
        else:
            raise ValueError("instruction %r is outside the simulated subset" % ln)
    return m.r
