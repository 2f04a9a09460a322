#!/usr/bin/env python3
"""Generator of the HAND-SCHEDULED gfx950 ladder bodies: DBL and ADD of a table entry, the DAG of
curve4q.py:138-171 on the signed radix-2^26 limbs of fp127.hip.h, each as ONE monolithic inline-asm body.

What the compiler does not get to decide here:
  * register allocation (temporaries live in a fixed physical range that the asm statement clobbers; a linear scan over the
    straight-line body assigns them),
  * the schedule (program order IS issue order: the real and imaginary accumulation chains of a product alternate, the carry
    of column k is the addend of column k+1's first multiply-add, operand preparation sits right in front of its product),
  * no opaque fences, hence no hazard s_nop, and no v_accvgpr moves.
Operand roles are chosen so that every negated imaginary part and every times-8 operand is computed once:
  DBL level 2 + R1toR3:  X = G*E, Z = G*F, Y = D*F, T = D*E   -> -G.im, -D.im, 8E, 8F           (26 ops for four products)
  ADD level 2:           X = E*F, Z = G*F, Y = G*H            -> -E.im, -G.im, 8F, 8H           (26 ops for three)

    python tools/asmgen/gen_ladder_step.py > fourq_amd/csrc/ladder_asm_gfx950.inc      (--stats: the instruction census)

Bodies and their operand order (every Fe2 is ten operands: re limbs 0-4, im limbs 0-4):
  FQ_ASM_DBL    X, Y, Z in/out | limb mask                                  R1/R4 -> (X, Y, Z) of the double
  FQ_ASM_DBLT   X, Y, Z in/out | T out | limb mask                          the same plus T = Ta*Tb of the double (R1toR3)
  FQ_ASM_ADD    X, Y, Z in/out | Ta out | Tb out | T in | N, D, E, F in | neg mask | limb mask       ADD_core with +-entry
  FQ_ASM_STEP   X, Y, Z in/out | Ta out | Tb out | N, D, E, F in | neg mask | limb mask              DBLT + ADD in one body
  FQ_ASM_MULU / SQRU   one product / square of the unsigned flavour (see body_mulu / body_sqru)
  FQ_ASM_TAU / UPSILON / CHI    X, Y, Z in/out | limb mask                  the endomorphisms' pieces, in place (curve4q.py:258-316)
  FQ_ASM_TAUDUAL  X, Y, Z in/out | N3, D3, F3 out | limb mask                tau_dual and R1toR3 of the result (E3 is Z)
  FQ_ASM_R1TOR2   N, D, E, F out | X, Y, Z, Ta, Tb in | limb mask             a tight, non-negative table entry
  FQ_ASM_TABLEADD N, D, E, F of the entry in/out | N3, D3, E3, F3 in | limb mask     entry <- R1toR2(ADD_core(p, entry))
"""
import argparse
import collections
import sys

POOL_LO, POOL_HI = 168, 255            # physical VGPRs every body clobbers (peak need: 84); v0-v167 stay the compiler's


class V:                                # one 32-bit value
    __slots__ = ("name", "fixed", "phys", "pair", "half")

    def __init__(self, name, fixed=None):
        self.name, self.fixed, self.phys, self.pair, self.half = name, fixed, None, None, 0

    def __repr__(self):
        return self.name


class P:                                # an aligned 64-bit pair
    def __init__(self, name):
        self.name = name
        self.lo, self.hi = V(name + ".lo"), V(name + ".hi")
        self.lo.pair = self.hi.pair = self
        self.hi.half = 1
        self.phys = None


class Prog:
    def __init__(self):
        self.ins = []                   # (mnemonic, template, defs, uses)
        self.n = 0
        self.signed = True              # False: the unsigned flavour of fp127.hip.h (v_mad_u64_u32, logical carry shifts)

    def v(self, name="t"):
        self.n += 1
        return V("%s%d" % (name, self.n))

    def p(self, name="p"):
        self.n += 1
        return P("%s%d" % (name, self.n))

    def vec(self, name, k=5):
        return [self.v(name) for _ in range(k)]

    def emit(self, mn, fmt, defs, uses):
        self.ins.append((mn, fmt, list(defs), list(uses)))

    # ---- instruction helpers (operands: V, P, or int literal) ----
    def op2(self, mn, d, a, b):         # d = a <op> b   (VOP2, e32 when a is a register/constant and b a VGPR)
        self.emit(mn, "{mn} {d}, {a}, {b}", [d], [x for x in (a, b) if isinstance(x, V)] + [("fmt", d, a, b)])

    def add(self, d, a, b):
        self.op2("v_add_u32_e32", d, a, b)

    def sub(self, d, a, b):             # d = a - b
        self.op2("v_sub_u32_e32", d, a, b)

    def shl(self, d, k, a):
        self.op2("v_lshlrev_b32_e32", d, k, a)

    def band(self, d, m, a):
        self.op2("v_and_b32_e32", d, m, a)

    def xor(self, d, a, b):
        self.op2("v_xor_b32_e32", d, a, b)

    def sel(self, d, m, x, y):          # d = m ? x : y  bitwise
        self.emit("v_bitop3_b32", "v_bitop3_b32 {0}, {1}, {2}, {3} bitop3:0xca", [d], [m, x, y])

    def mad(self, acc_out, a, b, acc_in):   # 64-bit acc_out = a*b + acc_in (signed); acc_in None -> 0
        uses = [a, b] + ([acc_in.lo, acc_in.hi] if acc_in is not None else [])
        self.emit("v_mad_i64_i32", "mad", [acc_out.lo, acc_out.hi], uses + [("mad", acc_out, a, b, acc_in)])

    def ashr64(self, d, k, s):
        self.emit("v_ashrrev_i64", "ashr", [d.lo, d.hi], [s.lo, s.hi, ("ashr", d, k, s)])

    def lshl_add64(self, d, s, k, c):   # d = (s << k) + c
        self.emit("v_lshl_add_u64", "lshladd", [d.lo, d.hi], [s.lo, s.hi, c.lo, c.hi, ("lshladd", d, s, k, c)])

    def alignbit(self, d, hi, lo, k):
        self.emit("v_alignbit_b32", "v_alignbit_b32 {0}, {1}, {2}, %d" % k, [d], [hi, lo])

    def mov0(self, d):
        self.emit("v_mov_b32_e32", "v_mov_b32_e32 {0}, 0", [d], [])

    def movk(self, d, value):           # a 32-bit literal (8 bytes)
        self.emit("v_mov_b32_e32", "v_mov_b32_e32 {0}, 0x%x" % value, [d], [])

    def addk(self, d, value, a):        # d = literal + a
        self.emit("v_add_u32_e32", "v_add_u32_e32 {0}, 0x%x, {1}" % value, [d], [a])

    def shr(self, d, k, a):             # logical
        self.op2("v_lshrrev_b32_e32", d, k, a)

    def shl_add(self, d, a, k, c):      # d = (a << k) + c
        self.emit("v_lshl_add_u32", "v_lshl_add_u32 {0}, {1}, %d, {2}" % k, [d], [a, c])


class Fe2:
    def __init__(self, re, im):
        self.re, self.im = re, im


def fixed_fe2(base):
    return Fe2([V("op%d" % (base + i), fixed="%%%d" % (base + i)) for i in range(5)],
               [V("op%d" % (base + 5 + i), fixed="%%%d" % (base + 5 + i)) for i in range(5)])


class Body:
    """One asm body.  `spec` lists the operands in order: (kind, attribute name) with kind 'fe2' (ten operands) or 'u32'."""

    def __init__(self, spec, pairs=False, order="re-im", seq=False, align=1):
        """pairs: the two products of a formula level run side by side (four accumulation chains instead of two);
        order: how a column's four multiply-adds per limb are laid out; seq: a column's real chain, then its imaginary chain."""
        self.g = Prog()
        self.pairs, self.order, self.seq, self.align = pairs, order, seq, align
        g = self.g
        n = 0
        for kind, name in spec:
            if kind == "fe2":
                setattr(self, name, fixed_fe2(n))
                n += 10
            else:
                setattr(self, name, V(name, fixed="%%%d" % n))
                n += 1
        self.n_operands = n
        # two pairs whose high halves stay zero: the zero-extended limb 0 of a product's two components
        self.zr, self.zi = g.p("zr"), g.p("zi")
        self.zsets = [(self.zr, self.zi)]
        if pairs:
            self.zsets.append((g.p("zr"), g.p("zi")))
        for zr, zi in self.zsets:
            zr.permanent = zi.permanent = True
            g.mov0(zr.hi)
            g.mov0(zi.hi)
        self.zsel = 0

    def side_by_side(self, first, second):
        """Emit two independent products; with `pairs` their instruction streams alternate one for one."""
        a0 = len(self.g.ins)
        self.zsel = 0
        r1 = first()
        a1 = len(self.g.ins)
        self.zsel = 1 if self.pairs else 0
        r2 = second()
        a2 = len(self.g.ins)
        self.zsel = 0
        if self.pairs:
            A, B = self.g.ins[a0:a1], self.g.ins[a1:a2]
            merged = []
            for k in range(max(len(A), len(B))):
                if k < len(A):
                    merged.append(A[k])
                if k < len(B):
                    merged.append(B[k])
            self.g.ins[a0:a2] = merged
        return r1, r2

    # ---- field helpers ----
    def fe2_new(self, name):
        return Fe2(self.g.vec(name + "r"), self.g.vec(name + "i"))

    def fe2_add(self, a, b, name="s", out=None):
        r = out or self.fe2_new(name)
        for i in range(5):
            self.g.add(r.re[i], a.re[i], b.re[i])
        for i in range(5):
            self.g.add(r.im[i], a.im[i], b.im[i])
        return r

    def fe2_sub(self, a, b, name="d", out=None):
        r = out or self.fe2_new(name)
        for i in range(5):
            self.g.sub(r.re[i], a.re[i], b.re[i])
        for i in range(5):
            self.g.sub(r.im[i], a.im[i], b.im[i])
        return r

    def neg_im(self, a):                # -a.im  (5 ops)
        r = self.g.vec("n")
        for i in range(5):
            self.g.sub(r[i], 0, a.im[i])
        return r

    def times8(self, a):                # (8 a.re[1..4], 8 a.im[1..4])   (8 ops); index 0 unused
        r8, i8 = [None] + self.g.vec("e", 4), [None] + self.g.vec("e", 4)
        for j in range(1, 5):
            self.g.shl(r8[j], 3, a.re[j])
        for j in range(1, 5):
            self.g.shl(i8[j], 3, a.im[j])
        return r8, i8

    # ---- the two accumulation chains of one product ----
    def columns(self, terms_re, terms_im, out):
        """terms_*[K] = list of (a, b) multiplicand pairs of column K.  Emits the chains with the real and imaginary parts
        alternating, the carry pass and the fold of the top carry (2^130 == 8).  Writes the ten limbs of `out`."""
        g = self.g
        acc_r, acc_i = g.p("ar"), g.p("ai")
        zr, zi = self.zsets[self.zsel]
        lr = [zr.lo] + [out.re[k] for k in range(1, 5)]
        li = [zi.lo] + [out.im[k] for k in range(1, 5)]
        first = True
        for K in range(5):
            tr, ti = terms_re[K], terms_im[K]
            if self.seq:
                for n in range(len(tr)):
                    g.mad(acc_r, tr[n][0], tr[n][1], None if (first and n == 0) else acc_r)
                for n in range(len(ti)):
                    g.mad(acc_i, ti[n][0], ti[n][1], None if (first and n == 0) else acc_i)
            else:
                for n in range(max(len(tr), len(ti))):
                    if n < len(tr):
                        g.mad(acc_r, tr[n][0], tr[n][1], None if (first and n == 0) else acc_r)
                    if n < len(ti):
                        g.mad(acc_i, ti[n][0], ti[n][1], None if (first and n == 0) else acc_i)
            first = False
            g.band(lr[K], self.mask, acc_r.lo)
            g.ashr64(acc_r, 26, acc_r)
            g.band(li[K], self.mask, acc_i.lo)
            g.ashr64(acc_i, 26, acc_i)
        # w = 8 * top + limb0 ; limb0' = w & mask ; limb1 += w >> 26
        for acc, z, o in ((acc_r, zr, out.re), (acc_i, zi, out.im)):
            g.lshl_add64(acc, acc, 3, z)
        for acc, z, o in ((acc_r, zr, out.re), (acc_i, zi, out.im)):
            g.band(o[0], self.mask, acc.lo)
            t = g.v("c")
            g.alignbit(t, acc.hi, acc.lo, 26)
            g.add(o[1], o[1], t)

    def mul(self, a, na_im, b, b8, name="m", out=None):
        """(a0 + a1 i)(b0 + b1 i): re = a0*b0 + (-a1)*b1, im = a0*b1 + a1*b0; wrap-around terms through 8b."""
        out = out or self.fe2_new(name)
        b0x8, b1x8 = b8
        tr, ti = [], []
        for K in range(5):
            r, m = [], []
            for i in range(5):
                j = K - i
                q0 = b.re[j] if j >= 0 else b0x8[j + 5]
                q1 = b.im[j] if j >= 0 else b1x8[j + 5]
                if self.order == "share-q":                    # neighbours share the second operand
                    r += [(a.re[i], q0), (na_im[i], q1)]
                    m += [(a.im[i], q0), (a.re[i], q1)]
                else:                                          # neighbours share the first operand
                    r += [(a.re[i], q0), (na_im[i], q1)]
                    m += [(a.re[i], q1), (a.im[i], q0)]
            tr.append(r)
            ti.append(m)
        self.columns(tr, ti, out)
        return out

    def sqr(self, a, wide=False, name="q", out=None):
        """(a0 + a1 i)^2: re = (a0 - a1)(a0 + a1), im = (2 a0) a1.  wide: 8(a0+a1) would not fit a signed operand, the
        wrap-around factor is split as (4d)(2s) (wrap_operands_signed of fp127.hip.h)."""
        g = self.g
        out = out or self.fe2_new(name)
        s, d, t = g.vec("s"), g.vec("d"), g.vec("t")
        for i in range(5):
            g.add(s[i], a.re[i], a.im[i])
        for i in range(5):
            g.sub(d[i], a.re[i], a.im[i])
        for i in range(5):
            g.shl(t[i], 1, a.re[i])
        sw, i8 = [None] + g.vec("e", 4), [None] + g.vec("e", 4)
        dw = d
        if wide:
            dw = [None] + g.vec("w", 4)
            for i in range(1, 5):
                g.shl(dw[i], 2, d[i])
        for j in range(1, 5):
            g.shl(sw[j], 1 if wide else 3, s[j])
        for j in range(1, 5):
            g.shl(i8[j], 3, a.im[j])
        tr, ti = [], []
        for K in range(5):
            r, m = [], []
            for i in range(5):
                j = K - i
                r.append((d[i], s[j]) if j >= 0 else (dw[i], sw[j + 5]))
                m.append((t[i], a.im[j]) if j >= 0 else (t[i], i8[j + 5]))
            tr.append(r)
            ti.append(m)
        self.columns(tr, ti, out)
        return out

    # ---- the formulas ----
    def dbl(self, with_t, out_t=None):
        """DBL (curve4q.py:138-152) of (X, Y, Z) in place; with_t: also T = E*D = Ta*Tb of the double (R1toR3, :119-126)."""
        X, Y, Z = self.X, self.Y, self.Z
        XY = self.fe2_add(X, Y, "xy")
        A, B = self.side_by_side(lambda: self.sqr(X, name="A"), lambda: self.sqr(Y, name="B"))
        Zs, W = self.side_by_side(lambda: self.sqr(Z, name="Zs"), lambda: self.sqr(XY, wide=True, name="W"))
        D = self.fe2_add(A, B, "D")
        E = self.fe2_sub(W, D, "E")
        F = self.fe2_sub(B, A, "F")
        C = self.fe2_add(Zs, Zs, "C")
        G = self.fe2_sub(C, F, "G")
        nG, nD = self.neg_im(G), self.neg_im(D)
        E8, F8 = self.times8(E), self.times8(F)
        T = None
        if with_t:
            T, _ = self.side_by_side(lambda: self.mul(D, nD, E, E8, "T", out=out_t), lambda: self.mul(G, nG, E, E8, out=X))
        else:
            self.mul(G, nG, E, E8, out=X)
        self.side_by_side(lambda: self.mul(D, nD, F, F8, out=Y), lambda: self.mul(G, nG, F, F8, out=Z))
        return T

    def add(self, T):
        """ADD_core (curve4q.py:155-175) of (X, Y, Z, T) and +-entry (N, D, E, F): N and D exchanged by masked selects
        (GFp2.select, fields.py:236-238), -F by a conditional negation; X, Y, Z in place, Ta = E, Tb = H."""
        g = self.g
        X2, Y2, Z2 = self.X, self.Y, self.Z
        N1 = self.fe2_add(X2, Y2, "N1")
        D1 = self.fe2_sub(Y2, X2, "D1")
        sN, sD, Fs = self.fe2_new("sN"), self.fe2_new("sD"), self.fe2_new("Fs")
        for part in ("re", "im"):
            for i in range(5):
                g.sel(getattr(sN, part)[i], self.neg, getattr(self.tD, part)[i], getattr(self.tN, part)[i])
            for i in range(5):
                g.sel(getattr(sD, part)[i], self.neg, getattr(self.tN, part)[i], getattr(self.tD, part)[i])
        for part in ("re", "im"):
            for i in range(5):
                g.xor(getattr(Fs, part)[i], self.neg, getattr(self.tF, part)[i])
            for i in range(5):
                g.sub(getattr(Fs, part)[i], getattr(Fs, part)[i], self.neg)
        A2, B2 = self.side_by_side(lambda: self.mul(D1, self.neg_im(D1), sD, self.times8(sD), "A2"),
                                   lambda: self.mul(N1, self.neg_im(N1), sN, self.times8(sN), "B2"))
        C2, D2 = self.side_by_side(lambda: self.mul(Fs, self.neg_im(Fs), T, self.times8(T), "C2"),
                                   lambda: self.mul(self.tE, self.neg_im(self.tE), Z2, self.times8(Z2), "D2"))
        E2 = self.fe2_sub(B2, A2, out=self.Ta)
        H2 = self.fe2_add(B2, A2, out=self.Tb)
        F2 = self.fe2_sub(D2, C2, "F2")
        G2 = self.fe2_add(D2, C2, "G2")
        nE2, nG2 = self.neg_im(E2), self.neg_im(G2)
        F28, H28 = self.times8(F2), self.times8(H2)
        self.side_by_side(lambda: self.mul(E2, nE2, F2, F28, out=self.X), lambda: self.mul(G2, nG2, H2, H28, out=self.Y))
        self.mul(G2, nG2, F2, F28, out=self.Z)


    # ---- table construction (table_endo, curve4q.py:385-403): the endomorphisms and the additions as whole-formula bodies ----
    UNIT = (1 << 26) + (1 << 15)

    def kmul(self, a, c, name="k", out=None):
        """a * constant c = (c0, c1) (ints mod p): the constant is the second operand, its limbs and their eightfold come as
        literals (one v_mov each) and every multiply-add with a ZERO limb of the constant is left out -- most of the curve's
        constants have three zero limbs in one component."""
        g = self.g
        out = out or self.fe2_new(name)
        lim = [[(c[k] >> (26 * i)) & ((1 << 26) - 1) for i in range(5)] for k in (0, 1)]
        regs = {}

        def lit(value):
            if value == 0:
                return None
            if value not in regs:
                regs[value] = g.v("c")
                g.movk(regs[value], value)
            return regs[value]
        na = self.neg_im(a)
        tr, ti = [], []
        for K in range(5):
            r, m = [], []
            for i in range(5):
                j = K - i
                w = 1 if j >= 0 else 8
                q0, q1 = lit(lim[0][j % 5] * w), lit(lim[1][j % 5] * w)
                if q0 is not None:
                    r.append((a.re[i], q0))
                    m.append((a.im[i], q0))
                if q1 is not None:
                    r.append((na[i], q1))
                    m.append((a.re[i], q1))
            tr.append(r)
            ti.append(m)
        self.columns(tr, ti, out)
        return out

    def fe2_neg(self, a, name="ng"):
        r = self.fe2_new(name)
        for part in ("re", "im"):
            for i in range(5):
                self.g.sub(getattr(r, part)[i], 0, getattr(a, part)[i])
        return r

    def fe2_conj(self, a, name="cj"):       # negates the imaginary part only; the real limbs are shared
        return Fe2(a.re, self.neg_im(a))

    def tighten(self, a, bound, out=None, name="tg"):
        """Signed limbs of magnitude <= bound * UNIT -> the same residue as NON-NEGATIVE tight limbs (what a table entry is for every
        consumer, fe_unsign of fp127.hip.h): add (bound + 1) * (2^130 - 8) in limb form, one 32-bit carry pass."""
        g = self.g
        out = out or self.fe2_new(name)
        M26 = (1 << 26) - 1
        for src, dst in ((a.re, out.re), (a.im, out.im)):
            u = g.vec("u")
            for i in range(5):
                g.addk(u[i], (bound + 1) * (M26 - 7 if i == 0 else M26), src[i])
            t = [None] + g.vec("t", 4)
            c = g.v("c")
            g.shr(c, 26, u[0]); g.add(t[1], u[1], c)
            for i in (2, 3, 4):
                c = g.v("c")
                g.shr(c, 26, t[i - 1]); g.add(t[i], u[i], c)
            l0, c = g.v("l"), g.v("c")
            g.band(l0, self.mask, u[0])
            g.shr(c, 26, t[4])
            t0 = g.v("t")
            g.shl_add(t0, c, 3, l0)                      # < 2^26 + 2^9
            g.band(dst[0], self.mask, t0)
            c2, l1 = g.v("c"), g.v("l")
            g.shr(c2, 26, t0)
            g.band(l1, self.mask, t[1])
            g.add(dst[1], l1, c2)
            for i in (2, 3, 4):
                g.band(dst[i], self.mask, t[i])
        return out

    def mulp(self, a, b, name="m", out=None):
        """a * b with the preparation done here (no sharing): b must be the operand whose eightfold fits (bound <= 3)."""
        return self.mul(a, self.neg_im(a), b, self.times8(b), name, out=out)

    def f_tau(self, X, Y, Z, oX, oY, oZ, ctau):                            # curve4q.py:258-267
        A, B = self.side_by_side(lambda: self.sqr(X, name="A"), lambda: self.sqr(Y, name="B"))
        C = self.fe2_add(A, B, "C")                       # 2
        D = self.fe2_sub(A, B, "D")                       # 2
        Zs = self.sqr(Z, name="Zs")
        t = self.fe2_add(self.fe2_add(Zs, Zs, "z2"), D, "t")          # 4
        nD = self.neg_im(D)
        C8 = self.times8(C)
        x1 = self.kmul(X, ctau, "x1")
        x2 = self.mulp(Y, x1, "x2")
        self.mul(D, nD, x2, self.times8(x2), out=oX)
        y = self.mul(t, self.neg_im(t), C, C8, "y")
        for part in ("re", "im"):
            for i in range(5):
                self.g.sub(getattr(oY, part)[i], 0, getattr(y, part)[i])
        self.mul(D, nD, C, C8, out=oZ)

    def f_tau_dual(self, X, Y, Z, oX, oY, oZ, oN3, oD3, oF3, ctaudual):     # curve4q.py:269-280 + R1toR3 (:119-126)
        A, B = self.side_by_side(lambda: self.sqr(X, name="A"), lambda: self.sqr(Y, name="B"))
        C = self.fe2_add(A, B, "C")                       # 2
        Ta = self.fe2_sub(B, A, "Ta")                     # 2
        Zs = self.sqr(Z, name="Zs")
        D = self.fe2_sub(self.fe2_add(Zs, Zs, "z2"), Ta, "D")         # 4
        tb1 = self.kmul(X, ctaudual, "tb1")
        Tb = self.mulp(Y, tb1, "Tb")
        nD = self.neg_im(D)
        Tb8 = self.times8(Tb)
        self.mul(C, self.neg_im(C), Tb, Tb8, out=oX)
        self.mul(D, nD, Ta, self.times8(Ta), out=oY)
        self.mul(D, nD, C, self.times8(C), out=oZ)
        self.mul(Ta, self.neg_im(Ta), Tb, Tb8, out=oF3)   # R1toR3: Ta * Tb
        self.fe2_add(oX, oY, out=oN3)
        self.fe2_sub(oY, oX, out=oD3)

    def f_upsilon(self, X, Y, Z, oX, oY, oZ, cphi):                          # curve4q.py:282-302
        a1 = self.kmul(X, cphi[0], "a1")
        A = self.mulp(Y, a1, "A")
        B = self.mulp(Y, Z, "B")
        C, D = self.side_by_side(lambda: self.sqr(Y, name="C"), lambda: self.sqr(Z, name="D"))
        F, G = self.side_by_side(lambda: self.sqr(D, name="F"), lambda: self.sqr(B, name="G"))
        H = self.sqr(C, name="H")
        I = self.kmul(B, cphi[1], "I")
        J = self.fe2_add(C, self.kmul(D, cphi[2], "j"), "J")                                        # 2
        K = self.fe2_add(self.fe2_add(self.kmul(G, cphi[8], "k8"), H, "k"), self.kmul(F, cphi[9], "k9"), "K")    # 3
        imj, ipj = self.fe2_sub(I, J, "imj"), self.fe2_add(I, J, "ipj")                              # 3, 3
        x2 = self.mulp(imj, ipj, "x2")
        L = self.fe2_add(C, self.kmul(D, cphi[4], "l"), "L")                                        # 2
        Mm = self.kmul(B, cphi[3], "M")
        lmm, lpm = self.fe2_sub(L, Mm, "lmm"), self.fe2_add(L, Mm, "lpm")                            # 3, 3
        Nn = self.mulp(lmm, lpm, "Nn")
        y2 = self.fe2_add(self.fe2_add(H, self.kmul(G, cphi[6], "y6"), "y"), self.kmul(F, cphi[7], "y7"), "y2")  # 3
        nK = self.neg_im(K)
        ka = self.mul(K, nK, A, self.times8(A), "ka")
        self.mulp(ka, x2, out=oX)
        d5 = self.kmul(D, cphi[5], "d5")
        dn = self.mulp(d5, Nn, "dn")
        self.mulp(y2, dn, out=oY)
        kb = self.mul(K, nK, B, self.times8(B), "kb")
        self.mulp(kb, Nn, out=oZ)
        for o in (oX, oY, oZ):                            # conj: the imaginary part negated in place
            for i in range(5):
                self.g.sub(o.im[i], 0, o.im[i])

    def f_chi(self, X, Y, Z, oX, oY, oZ, cpsi):                                # curve4q.py:304-316
        A, B, Zc = self.fe2_conj(X), self.fe2_conj(Y), self.fe2_conj(Z)
        C, D = self.side_by_side(lambda: self.sqr(Zc, name="C"), lambda: self.sqr(A, name="D"))
        g1 = self.fe2_add(D, self.kmul(C, cpsi[2], "c2"), "g1")       # 2
        G = self.mulp(B, g1, "G")
        h1 = self.fe2_add(D, self.kmul(C, cpsi[4], "c4"), "h1")       # 2
        H = self.fe2_neg(h1, "H")
        nH = self.neg_im(H)
        ac = self.mulp(self.kmul(A, cpsi[1], "a1"), C, "ac")
        self.mul(H, nH, ac, self.times8(ac), out=oX)
        y1 = self.fe2_add(D, self.kmul(C, cpsi[3], "c3"), "y1")       # 2
        G8 = self.times8(G)
        self.mul(y1, self.neg_im(y1), G, G8, out=oY)
        self.mul(H, nH, G, G8, out=oZ)

    def f_r1_to_r2(self, X, Y, Z, Ta, Tb, oN, oD, oE, oF, two_d, in_bound=1):    # curve4q.py:109-116; tight non-negative entry
        self.tighten(self.fe2_add(X, Y, "n"), 2 * in_bound, out=oN)
        self.tighten(self.fe2_sub(Y, X, "d"), 2 * in_bound, out=oD)
        self.tighten(self.fe2_add(Z, Z, "e"), 2 * in_bound, out=oE)
        tt = self.mulp(Ta, Tb, "tt")
        self.tighten(self.kmul(tt, two_d, "f"), 1, out=oF)

    def f_table_add(self, pN, pD, pE, pF, qN, qD, qE, qF, oN, oD, oE, oF, two_d):   # r1_to_r2(add_core(p, q)), curve4q.py:155-171, :109-116
        A, B = self.side_by_side(lambda: self.mulp(pD, qD, "A"), lambda: self.mulp(pN, qN, "B"))
        C, D = self.side_by_side(lambda: self.mulp(qF, pF, "C"), lambda: self.mulp(qE, pE, "D"))
        E = self.fe2_sub(B, A, "E")
        F = self.fe2_sub(D, C, "F")
        G = self.fe2_add(D, C, "G")
        H = self.fe2_add(B, A, "H")
        nE, nG = self.neg_im(E), self.neg_im(G)
        F8, H8 = self.times8(F), self.times8(H)
        X, Y = self.side_by_side(lambda: self.mul(E, nE, F, F8, "X"), lambda: self.mul(G, nG, H, H8, "Y"))
        Z, tt = self.side_by_side(lambda: self.mul(G, nG, F, F8, "Z"), lambda: self.mul(E, nE, H, H8, "tt"))
        self.tighten(self.fe2_add(X, Y, "n"), 2, out=oN)
        self.tighten(self.fe2_sub(Y, X, "d"), 2, out=oD)
        self.tighten(self.fe2_add(Z, Z, "e"), 2, out=oE)
        self.tighten(self.kmul(tt, two_d, "f"), 1, out=oF)


XYZ = [("fe2", "X"), ("fe2", "Y"), ("fe2", "Z")]
ENTRY = [("fe2", "tN"), ("fe2", "tD"), ("fe2", "tE"), ("fe2", "tF")]


def body_dbl():
    b = Body(XYZ + [("u32", "mask")])
    b.dbl(False)
    return b


def body_dblt():
    b = Body(XYZ + [("fe2", "T"), ("u32", "mask")])
    b.dbl(True, out_t=b.T)
    return b


def body_add():
    b = Body(XYZ + [("fe2", "Ta"), ("fe2", "Tb"), ("fe2", "T")] + ENTRY + [("u32", "neg"), ("u32", "mask")])
    b.add(b.T)
    return b


def body_step(**options):
    b = Body(XYZ + [("fe2", "Ta"), ("fe2", "Tb")] + ENTRY + [("u32", "neg"), ("u32", "mask")], **options)
    b.add(b.dbl(True))
    return b


# experimental step bodies for tools/microbench/ladder_step.hip (--variants): what the order of the same instructions is worth
VARIANTS = [("STEP_V3", "H as the assembler lays it out (no placement)", dict(align=0)),
            ("STEP_V4", "H with every run of multiply-adds at 4 mod 8", dict(align=2)),
            ("STEP_V5", "two products side by side (four accumulation chains)", dict(pairs=True)),
            ("STEP_V6", "a column's real chain, then its imaginary chain (dependent neighbours)", dict(seq=True))]


def operand_vec(name, base, k=5):
    return [V("%s%d" % (name, i), fixed="%%%d" % (base + i)) for i in range(k)]


def body_mulu():
    """One GF(p^2) product of the UNSIGNED flavour (fe2_mul_plain of fp127.hip.h: non-negative lazy limbs, the caller passes the
    biased negation of a.im): operands out | a.re a.im (-a.im) | b | mask.  No placement of its own: the build's placement pass
    (tools/asmgen/place_asm.py) sees these instructions like hipcc's."""
    b = Body([("fe2", "C"), ("fe2", "A")], align=0)
    b.g.signed = False
    na = operand_vec("na", 20)
    B = fixed_fe2(25)
    b.mask = V("mask", fixed="%35")
    b.n_operands = 36
    b.mul(b.A, na, B, b.times8(B), out=b.C)
    return b


def body_sqru():
    """One GF(p^2) square of the unsigned flavour as its two GF(p) products re = d * s, im = t * a.im (d = a.re - a.im biased,
    s = a.re + a.im, t = 2 a.re: computed by the caller): operands out | d | s | t | a.im | mask."""
    b = Body([("fe2", "C")], align=0)
    b.g.signed = False
    d, s, t, im = (operand_vec(n, 10 + 5 * k) for k, n in enumerate(("d", "s", "t", "im")))
    b.mask = V("mask", fixed="%30")
    b.n_operands = 31
    g = b.g
    s8, i8 = [None] + g.vec("e", 4), [None] + g.vec("e", 4)
    for j in range(1, 5):
        g.shl(s8[j], 3, s[j])
    for j in range(1, 5):
        g.shl(i8[j], 3, im[j])
    tr, ti = [], []
    for K in range(5):
        r, m = [], []
        for i in range(5):
            j = K - i
            r.append((d[i], s[j]) if j >= 0 else (d[i], s8[j + 5]))
            m.append((t[i], im[j]) if j >= 0 else (t[i], i8[j + 5]))
        tr.append(r)
        ti.append(m)
    b.columns(tr, ti, b.C)
    return b


def _constants():
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("fq_constants", os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "fourq_amd", "constants.py"))
    K = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(K)
    return K


def _spec(*names):
    return [("fe2", n) for n in names]


# The table bodies work IN PLACE where a value's last read precedes the result's first write (every formula below writes its
# results with its last products / its closing negations), which keeps their operand lists short.
def body_tau():
    K = _constants()
    b = Body(_spec("X", "Y", "Z") + [("u32", "mask")])
    b.f_tau(b.X, b.Y, b.Z, b.X, b.Y, b.Z, K.ctau)
    return b


def body_taudual():
    K = _constants()
    b = Body(_spec("X", "Y", "Z", "oN3", "oD3", "oF3") + [("u32", "mask")])
    b.f_tau_dual(b.X, b.Y, b.Z, b.X, b.Y, b.Z, b.oN3, b.oD3, b.oF3, K.ctaudual)
    return b


def body_upsilon():
    K = _constants()
    b = Body(_spec("X", "Y", "Z") + [("u32", "mask")])
    b.f_upsilon(b.X, b.Y, b.Z, b.X, b.Y, b.Z, K.cphi)
    return b


def body_chi():
    K = _constants()
    b = Body(_spec("X", "Y", "Z") + [("u32", "mask")])
    b.f_chi(b.X, b.Y, b.Z, b.X, b.Y, b.Z, K.cpsi)
    return b


def _two_d(K):
    return ((2 * K.d[0]) % K.P127, (2 * K.d[1]) % K.P127)


def body_r1tor2():
    K = _constants()
    b = Body(_spec("oN", "oD", "oE", "oF", "X", "Y", "Z", "Ta", "Tb") + [("u32", "mask")])          # outputs first: the operand numbering of an asm statement
    b.f_r1_to_r2(b.X, b.Y, b.Z, b.Ta, b.Tb, b.oN, b.oD, b.oE, b.oF, _two_d(K))
    return b


def body_tableadd():
    K = _constants()
    b = Body(_spec("qN", "qD", "qE", "qF", "pN", "pD", "pE", "pF") + [("u32", "mask")])       # q (the table entry) in, the new entry out, in place
    b.f_table_add(b.pN, b.pD, b.pE, b.pF, b.qN, b.qD, b.qE, b.qF, b.qN, b.qD, b.qE, b.qF, _two_d(K))
    return b


TABLE_BODIES = ("TAU", "TAUDUAL", "UPSILON", "CHI", "R1TOR2", "TABLEADD")
SMALL_BODIES, SMALL_POOL_LO = ("MULU", "SQRU"), 236          # single products: 16 temporaries, their own short clobber list
BODIES = [("DBL", body_dbl), ("DBLT", body_dblt), ("ADD", body_add), ("STEP", body_step), ("MULU", body_mulu), ("SQRU", body_sqru),
          ("TAU", body_tau), ("TAUDUAL", body_taudual), ("UPSILON", body_upsilon), ("CHI", body_chi), ("R1TOR2", body_r1tor2), ("TABLEADD", body_tableadd)]


def allocate(prog, POOL_LO=POOL_LO, top_down=False):
    """Linear scan over the straight-line body: temporaries get physical registers POOL_LO..POOL_HI (pairs even-aligned,
    taken from the top; singles from the bottom).  A register freed by instruction i is reusable from instruction i+1."""
    def unit(x):                                          # the allocation unit of a value: itself, or its pair
        return x.pair if x.pair is not None else x

    last = {}
    for idx, (_, _, defs, uses) in enumerate(prog.ins):
        for x in list(defs) + [u for u in uses if isinstance(u, V)]:
            if not x.fixed:
                last[id(unit(x))] = idx
    free = set(range(POOL_LO, POOL_HI + 1))
    peak, pending = 0, []

    lowest = [POOL_HI + 1]

    def take_single():
        order = range(POOL_HI, POOL_LO - 1, -1) if top_down else range(POOL_LO, POOL_HI + 1)
        for r in order:
            if r in free:
                free.discard(r)
                lowest[0] = min(lowest[0], r)
                return r
        raise SystemExit("out of temporaries")

    def take_pair():
        for r in range(POOL_HI - 1, POOL_LO - 1, -2):
            if r in free and r + 1 in free:
                free.discard(r)
                free.discard(r + 1)
                lowest[0] = min(lowest[0], r)
                return r
        raise SystemExit("out of temporary pairs")

    for idx, (_, _, defs, uses) in enumerate(prog.ins):
        free.update(pending)
        pending = []
        for d in defs:
            if d.fixed:
                continue
            u = unit(d)
            if u.phys is None:
                if isinstance(u, P):
                    u.phys = take_pair()
                    u.lo.phys, u.hi.phys = u.phys, u.phys + 1
                else:
                    u.phys = take_single()
        peak = max(peak, (POOL_HI - POOL_LO + 1) - len(free))
        done = set()
        for x in list(defs) + [u for u in uses if isinstance(u, V)]:
            if x.fixed:
                continue
            u = unit(x)
            if id(u) in done or getattr(u, "permanent", False):
                continue
            done.add(id(u))
            if u.phys is None:
                raise SystemExit("use of %s before its definition (instruction %d)" % (x.name, idx))
            if last[id(u)] == idx:
                pending += [u.phys, u.phys + 1] if isinstance(u, P) else [u.phys]
    prog.lowest_register = lowest[0]
    return peak


def reg(x):
    if isinstance(x, int):
        return str(x)
    if x.fixed:
        return x.fixed
    return "v%d" % x.phys


def preg(p):
    return "v[%d:%d]" % (p.phys, p.phys + 1)


def render(prog):
    lines = []
    mad_mn = "v_mad_i64_i32" if prog.signed else "v_mad_u64_u32"
    shr_mn = "v_ashrrev_i64" if prog.signed else "v_lshrrev_b64"
    for mn, fmt, defs, uses in prog.ins:
        tag = [u for u in uses if isinstance(u, tuple)]
        if fmt == "mad":
            _, out, a, b, acc = tag[0]
            lines.append("%s %s, vcc, %s, %s, %s" % (mad_mn, preg(out), reg(a), reg(b), preg(acc) if acc is not None else "0"))
        elif fmt == "ashr":
            _, d, k, s = tag[0]
            lines.append("%s %s, %d, %s" % (shr_mn, preg(d), k, preg(s)))
        elif fmt == "lshladd":
            _, d, s, k, c = tag[0]
            lines.append("v_lshl_add_u64 %s, %s, %d, %s" % (preg(d), preg(s), k, preg(c)))
        elif tag and tag[0][0] == "fmt":
            _, d, a, b = tag[0]
            lines.append("%s %s, %s, %s" % (mn, reg(d), reg(a), reg(b)))
        else:
            ops = [reg(x) for x in defs] + [reg(u) for u in uses if isinstance(u, V)]
            lines.append(fmt.format(*ops))
    return lines


def place(lines, mode):
    """Code placement.  A VOP3 instruction is 8 bytes, an _e32 one 4: after an odd number of 4-byte instructions every following
    multiply-add straddles an 8-byte fetch boundary.  mode 1: the body starts 8-byte aligned and wherever an 8-byte instruction would
    start at 4 (mod 8) the nearest 4-byte instruction in front of it is emitted in its 8-byte (_e64) encoding instead -- same
    instruction count, no padding.  mode 2 (experiment): the opposite, every run of 8-byte instructions starts at 4 (mod 8)."""
    if not mode:
        return lines
    out, off, last32 = [".p2align 3"], 0, None
    for ln in lines:
        literal = " 0x" in ln                                   # a 32-bit literal rides behind the instruction: 8 bytes already, no _e64 form
        size = 4 if (ln.split()[0].endswith("_e32") and not literal) else 8
        if size == 8:
            want = 0 if mode == 1 else 4
            if off % 8 != want and last32 is not None:
                out[last32] = out[last32].replace("_e32", "_e64", 1)
                off += 4
            last32 = None
        else:
            last32 = len(out)
        out.append(ln)
        off += size
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--stats", action="store_true")
    ap.add_argument("--variants", action="store_true", help="emit the experimental step bodies instead of the product's")
    args = ap.parse_args()
    built = []
    bodies = BODIES if not args.variants else [(n, (lambda o=o: body_step(**o))) for n, _, o in VARIANTS]
    pool_lo = 128 if args.variants else POOL_LO              # the experiments may use a larger pool
    for name, make in bodies:
        b = make()
        if name in TABLE_BODIES:                                   # own clobber list, as short as the body's peak allows
            peak = allocate(b.g, 40, top_down=True)
        else:
            peak = allocate(b.g, SMALL_POOL_LO if name in SMALL_BODIES else pool_lo)
        lines = place(render(b.g), b.align)
        built.append((name, b, peak, lines, collections.Counter(ln.split()[0].replace('_e64', '_e32') for ln in lines if not ln.startswith('.'))))
    if args.stats:
        for name, b, peak, lines, census in built:
            lines = [ln for ln in lines if not ln.startswith(".")]
            print("%-8s %d instructions, %d operands, peak temporaries %d (lowest register v%s)" % (name, len(lines), b.n_operands, peak, getattr(b.g, "lowest_register", "?")))
            print("      " + "  ".join("%s %d" % kv for kv in census.most_common()))
        return
    out = sys.stdout
    out.write("// GENERATED by tools/asmgen/gen_ladder_step.py -- do not edit (tests/test_host.py checks that it is current).\n")
    out.write("// Hand-scheduled gfx950 bodies of the ladder's doubling and addition (curve4q.py:138-171); operand order in the generator's docstring.\n")
    out.write("// Temporaries: v%d-v%d, clobbered by every body.\n" % (pool_lo, POOL_HI))
    if args.variants:
        for n, text, _ in VARIANTS:
            out.write("#define FQ_%s_NAME \"%s\"\n" % (n, text))
    out.write("#define %s \"vcc\"" % ("FQ_ASM_VARIANT_CLOBBERS" if args.variants else "FQ_ASM_CLOBBERS"))
    for r in range(pool_lo, POOL_HI + 1):
        out.write(", \"v%d\"" % r)
    out.write("\n")
    if not args.variants:
        out.write("#define FQ_ASM_SMALL_CLOBBERS \"vcc\"")
        for r in range(SMALL_POOL_LO, POOL_HI + 1):
            out.write(", \"v%d\"" % r)
        out.write("\n")
    for name, b, peak, lines, census in built:
        if name in TABLE_BODIES:
            out.write("#define FQ_ASM_%s_CLOBBERS \"vcc\"" % name)
            for r in range(b.g.lowest_register, POOL_HI + 1):
                out.write(", \"v%d\"" % r)
            out.write("\n")
        out.write("// %s: %d instructions (%s), peak %d temporaries\n" % (name, len(lines), ", ".join("%d %s" % (v, k) for k, v in census.most_common()), peak))
        out.write("#define FQ_ASM_%s \\\n" % name)
        for ln in lines:
            out.write("    \"%s\\n\" \\\n" % ln)
        out.write("    \"\"\n")


if __name__ == "__main__":
    main()
