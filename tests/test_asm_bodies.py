"""The GENERATED gfx950 bodies (tools/asmgen/gen_ladder_step.py -> fourq_amd/csrc/ladder_asm_gfx950.inc) against the oracle's formulas, on
the CPU: tools/asmgen/sim.py interprets each body's instruction stream on Python integers for random inputs, the outputs' residues must be
the oracle's (DBL, ADD_core, R1toR2 / R1toR3, tau, tau_dual, upsilon, chi: curve4q.py:109-175, :258-316), table entries must come out as
tight non-negative limbs, and every 8-byte instruction must sit on an 8-byte boundary.  The GPU suite checks the same bodies end to end
(every MUL_* / DH_* output bit-exact); this is the part of that evidence a box without a GPU can reproduce."""
import os
import random
import re
import sys

import pytest

import curve4q_oracle as o
from conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, "tools", "asmgen"))
import gen_ladder_step as gen      # noqa: E402
import sim                         # noqa: E402

P = (1 << 127) - 1
M26 = (1 << 26) - 1


def limbs(x):
    return [(x >> (26 * i)) & M26 for i in range(5)]


def value(ls, signed=True):
    """residue of five 32-bit limbs read as signed (or unsigned) radix-2^26 digits"""
    tot = 0
    for i, l in enumerate(ls):
        if signed and l & (1 << 31):
            l -= 1 << 32
        tot += l << (26 * i)
    return tot % P


INC = os.path.join(ROOT, "fourq_amd", "csrc", "ladder_asm_gfx950.inc")


def _parse_inc():
    """the instruction lines and clobber lists of the SHIPPED include file (what the library is compiled from; test_host.py checks that
    the generator reproduces it byte for byte)"""
    bodies, clobbers, cur = {}, {}, None
    for ln in open(INC):
        m = re.match(r"#define FQ_ASM_(\w+?)(_CLOBBERS)? (.*)$", ln.rstrip("\n"))
        if m and (m.group(2) or m.group(1) == "CLOBBERS"):           # FQ_ASM_CLOBBERS (ladder bodies), FQ_ASM_SMALL_CLOBBERS, FQ_ASM_<body>_CLOBBERS
            clobbers[m.group(1) if m.group(2) else ""] = set(re.findall(r'"(v\d+)"', m.group(3)))
            cur = None
        elif m:
            cur = bodies.setdefault(m.group(1), [])
        elif cur is not None:
            t = re.match(r'\s*"(.*?)(?:\\n)?"', ln)
            if t and t.group(1):
                cur.append(t.group(1))
            if not ln.rstrip().endswith("\\"):
                cur = None
    return bodies, clobbers


BODY_TEXT, CLOBBERS = _parse_inc()


def body_lines(name):
    return BODY_TEXT[name]


def put_fe2(regs, base, v):
    for i, l in enumerate(limbs(v[0]) + limbs(v[1])):
        regs["%%%d" % (base + i)] = l


def get_fe2(regs, base, signed=True):
    ls = [regs["%%%d" % (base + i)] for i in range(10)]
    return (value(ls[:5], signed), value(ls[5:], signed))


def rnd_fe2(rng):
    return (rng.randrange(P), rng.randrange(P))


def rnd_point_r1(rng):
    """a random N-torsion point in R1 with Z != 1 (the ladder's running point)"""
    return o.MUL_endo(rng.getrandbits(200) | 1, o.AffineToR1(o.Gx, o.Gy))


@pytest.fixture(scope="module")
def rng():
    return random.Random(2604)


def test_dbl_and_dblt(rng):
    for name in ("DBL", "DBLT"):
        lines = body_lines(name)
        for _ in range(4):
            Q = rnd_point_r1(rng)
            regs = {}
            for k in range(3):
                put_fe2(regs, 10 * k, Q[k])
            regs["%%%d" % (40 if name == "DBLT" else 30)] = M26
            out = sim.run(lines, regs)
            want = o.DBL(Q)
            assert [get_fe2(out, 10 * k) for k in range(3)] == [want[0], want[1], want[2]], name
            if name == "DBLT":
                assert get_fe2(out, 30) == o.GFp2.mul(want[3], want[4])


@pytest.mark.parametrize("neg", [0, 0xFFFFFFFF])
def test_add_and_step(rng, neg):
    add, step = body_lines("ADD"), body_lines("STEP")
    for _ in range(3):
        Q, R = rnd_point_r1(rng), rnd_point_r1(rng)
        entry = o.R1toR2(R)
        chosen = o.R2neg(entry) if neg else entry
        # ADD: (X, Y, Z, T) + entry
        regs = {}
        for k in range(3):
            put_fe2(regs, 10 * k, Q[k])
        put_fe2(regs, 50, o.GFp2.mul(Q[3], Q[4]))
        for k in range(4):
            put_fe2(regs, 60 + 10 * k, entry[k])
        regs["%100"], regs["%101"] = neg, M26
        out = sim.run(add, regs)
        want = o.ADD_core(o.R1toR3(Q), chosen)
        assert [get_fe2(out, 10 * k) for k in range(5)] == list(want)
        # STEP: DBL then ADD
        regs = {}
        for k in range(3):
            put_fe2(regs, 10 * k, Q[k])
        for k in range(4):
            put_fe2(regs, 50 + 10 * k, entry[k])
        regs["%90"], regs["%91"] = neg, M26
        out = sim.run(step, regs)
        want = o.ADD(o.DBL(Q), chosen)
        assert [get_fe2(out, 10 * k) for k in range(5)] == list(want)


def test_unsigned_product_and_square(rng):
    mulu, sqru = body_lines("MULU"), body_lines("SQRU")
    for _ in range(6):
        a, b = rnd_fe2(rng), rnd_fe2(rng)
        regs = {}
        put_fe2(regs, 10, a)
        bias = [2 * (M26 - 7)] + [2 * M26] * 4                      # fe_neg of a bound-1 element: 2 * (2^130 - 8) in limb form, minus the limbs
        for i, l in enumerate(limbs(a[1])):
            regs["%%%d" % (20 + i)] = bias[i] - l
        put_fe2(regs, 25, b)
        regs["%35"] = M26
        out = sim.run(mulu, regs)
        assert get_fe2(out, 0, signed=False) == o.GFp2.mul(a, b)
        assert all(0 <= out["%%%d" % i] < (1 << 26) + (1 << 15) for i in range(10))
        s = [x + y for x, y in zip(limbs(a[0]), limbs(a[1]))]
        d = [x + (bb - y) for x, y, bb in zip(limbs(a[0]), limbs(a[1]), bias)]
        t = [2 * x for x in limbs(a[0])]
        regs = {"%30": M26}
        for i in range(5):
            regs["%%%d" % (10 + i)], regs["%%%d" % (15 + i)], regs["%%%d" % (20 + i)], regs["%%%d" % (25 + i)] = d[i], s[i], t[i], limbs(a[1])[i]
        out = sim.run(sqru, regs)
        assert get_fe2(out, 0, signed=False) == o.GFp2.sqr(a)


def _is_tight(regs, base):
    return all(0 <= regs["%%%d" % (base + i)] < (1 << 26) + (1 << 15) for i in range(10))


def test_table_formulas(rng):
    tau, ups, chi, td = body_lines("TAU"), body_lines("UPSILON"), body_lines("CHI"), body_lines("TAUDUAL")
    for _ in range(2):
        Q = rnd_point_r1(rng)
        regs = {"%30": M26}
        for k in range(3):
            put_fe2(regs, 10 * k, Q[k])
        out = sim.run(tau, regs)
        t = o.tau(Q[:3])
        assert [get_fe2(out, 10 * k) for k in range(3)] == list(t)
        for lines, fn in ((ups, o.upsilon), (chi, o.chi)):
            regs = {"%30": M26}
            for k in range(3):
                put_fe2(regs, 10 * k, t[k])
            out = sim.run(lines, regs)
            u = fn(t)
            assert [get_fe2(out, 10 * k) for k in range(3)] == list(u)[:3], fn.__name__
            regs = {"%60": M26}
            for k in range(3):
                put_fe2(regs, 10 * k, u[k])
            out2 = sim.run(td, regs)
            V = o.tau_dual(u)
            V3 = o.R1toR3(V)
            assert [get_fe2(out2, 10 * k) for k in range(3)] == [V[0], V[1], V[2]]
            assert (get_fe2(out2, 30), get_fe2(out2, 40), get_fe2(out2, 50)) == (V3[0], V3[1], V3[3])


def test_table_entries(rng):
    r1tor2, tadd = body_lines("R1TOR2"), body_lines("TABLEADD")
    for _ in range(3):
        Q, R = rnd_point_r1(rng), rnd_point_r1(rng)
        regs = {"%90": M26}
        for k in range(5):
            put_fe2(regs, 40 + 10 * k, Q[k])
        out = sim.run(r1tor2, regs)
        want = o.R1toR2(Q)
        assert [get_fe2(out, 10 * k, signed=False) for k in range(4)] == list(want)
        assert all(_is_tight(out, 10 * k) for k in range(4))
        entry, p3 = o.R1toR2(R), o.R1toR3(Q)
        regs = {"%80": M26}
        for k in range(4):
            put_fe2(regs, 10 * k, entry[k])
            put_fe2(regs, 40 + 10 * k, p3[k])
        out = sim.run(tadd, regs)
        want = o.R1toR2(o.ADD_core(p3, entry))
        assert [get_fe2(out, 10 * k, signed=False) for k in range(4)] == list(want)
        assert all(_is_tight(out, 10 * k) for k in range(4))


def test_every_wide_instruction_of_every_body_is_placed():
    """an 8-byte instruction that starts at 4 (mod 8) issues slower; the generator's placement must leave none in the
    bodies it places itself (the single products are placed by the build's pass instead)."""
    assert set(BODY_TEXT) == {n for n, _ in gen.BODIES}
    for name, _ in gen.BODIES:
        if name in gen.SMALL_BODIES:
            continue
        lines = body_lines(name)
        assert lines[0] == ".p2align 3"
        off = 0
        for ln in lines[1:]:
            mn = ln.split()[0]
            size = 4 if (mn.endswith("_e32") and " 0x" not in ln) else 8
            assert not (size == 8 and off % 8), (name, ln)
            off += size


def test_bodies_write_only_their_operands_and_the_registers_they_declare_clobbered():
    """a physical register a body writes outside its clobber list would silently corrupt a value the compiler keeps there"""
    for name, lines in BODY_TEXT.items():
        declared = CLOBBERS[name] if name in CLOBBERS else CLOBBERS["SMALL" if name in gen.SMALL_BODIES else ""]
        for ln in lines:
            if ln.startswith("."):
                continue
            dst = ln.split(None, 1)[1].split(",")[0].strip()
            m = re.fullmatch(r"v\[(\d+):(\d+)\]", dst)
            regs = ["v%d" % r for r in range(int(m.group(1)), int(m.group(2)) + 1)] if m else [dst]
            for r in regs:
                assert r.startswith("%") or r in declared, (name, ln)
