"""Mirror of the reference's field namespaces `GFp` / `GFp2` (impl/fields.py:9-238), computed by
the device field layer (fp127.hip.h) through the primitive ABI.  Static methods, tuples of ints in
and out, results canonical in [0, p) exactly as `% p1271` returns them."""
import numpy as np

from . import codec
from .constants import P127
from .engine import default_engine

p1271 = P127


def _fp(op, x, y=0):
    row = np.array([[*codec._fp_words(x), *codec._fp_words(y)]], dtype=np.uint64)
    out = default_engine().prim(op, row)[0]
    return int(out[0]) | (int(out[1]) << 64)


def _raw128(x):
    x = int(x)
    if x < 0 or x >> 128:
        raise ValueError("select takes values in [0, 2^128)")
    return (x & codec.M64, x >> 64)


def _select(op, c, parts_x, parts_y):
    """GFp.select / GFp2.select on the device: y ^ ((mask * c) & (x ^ y)) on raw 128-bit words (fields.py:59-64)."""
    words = [*_raw128(int(c) % (1 << 128))]
    for v in (*parts_x, *parts_y):
        words.extend(_raw128(v))
    out = default_engine().prim(op, np.array([words], dtype=np.uint64))[0]
    vals = [int(out[i]) | (int(out[i + 1]) << 64) for i in range(0, len(out), 2)]
    return vals


def _fp2(op, a, b=(0, 0)):
    row = codec.pack_fp2s([a, b]).reshape(1, 8)
    return codec.unpack_fp2s(default_engine().prim(op, row)[0])[0]


class GFp:
    # the reference's operation counters (fields.py:10-27): class attributes that its compare.py resets and reads.  Calls of the
    # methods below count exactly as the reference's do; the curve-level functions of fourq_amd.curve4q run as ONE device call
    # each and do not pass through these methods, so they leave the counters alone (tools/compare.py prints the reference's
    # per-function table from its own CPU restatement instead).
    ctr_enabled = True
    A = S = M = I = 0                                  # noqa: E741
    half = 1 << 126                                    # fields.py:16

    @staticmethod
    def ctr_reset():                                   # fields.py:18-23
        GFp.A = GFp.S = GFp.M = GFp.I = 0

    @staticmethod
    def ctr():                                         # fields.py:25-27 names a class that does not exist: kept, like the decode bug
        raise NameError("name 'GFp1271' is not defined")

    @staticmethod
    def _count(which, by=1):
        if GFp.ctr_enabled:
            setattr(GFp, which, getattr(GFp, which) + by)

    @staticmethod
    def add(x, y):
        GFp._count("A")                                     # fields.py:30
        return _fp("FP_ADD", x, y)

    @staticmethod
    def sub(x, y):
        GFp._count("A")                                     # fields.py:36
        return _fp("FP_SUB", x, y)

    @staticmethod
    def mul(x, y):
        GFp._count("M")                                     # fields.py:42
        return _fp("FP_MUL", x, y)

    @staticmethod
    def sqr(x):
        GFp._count("S")                                        # fields.py:48
        return _fp("FP_SQR", x)

    @staticmethod
    def neg(x):
        GFp._count("A")                                        # fields.py:54
        return _fp("FP_NEG", x)

    @staticmethod
    def inv(x):                                        # fields.py:67 (the reference's chain calls sqr 126 times and mul 12 times)
        GFp._count("S", 126)
        GFp._count("M", 12)
        return _fp("FP_INV", x)

    @staticmethod
    def invsqrt(x):                                    # fields.py:110 (7 + 24 mul, 120 sqr in the reference's chain)
        GFp._count("M", 31)
        GFp._count("S", 120)
        return _fp("FP_INVSQRT", x)

    @staticmethod
    def toLittleEndian(x):                             # fields.py:125
        return bytearray(int(x).to_bytes(16, "little"))

    @staticmethod
    def fromLittleEndian(x):                           # fields.py:129 (masks bit 127, in place like the reference)
        x[15] &= 0x7F
        return int.from_bytes(bytes(x), "little")

    @staticmethod
    def select(c, x, y):                               # fields.py:60 (x if c == 1, y if c == 0; masked XOR on the device)
        return _select("FP_SELECT", c, (x,), (y,))[0]


class GFp2:
    A = S = M = I = 0                                  # noqa: E741  (fields.py:135-138; counted unconditionally, as there)
    zero, one, two = (0, 0), (1, 0), (2, 0)            # fields.py:140-142
    half = (GFp.half, 0)

    @staticmethod
    def ctr_reset():                                   # fields.py:145-149
        GFp2.A = GFp2.S = GFp2.M = GFp2.I = 0

    @staticmethod
    def ctr():                                         # fields.py:152-153
        return (GFp2.A, GFp2.S, GFp2.M, GFp2.I)

    @staticmethod
    def add(a, b):                                     # fields.py:157
        GFp2.A += 1
        return _fp2("FP2_ADD", a, b)

    @staticmethod
    def sub(a, b):                                     # fields.py:162
        GFp2.A += 1
        return _fp2("FP2_SUB", a, b)

    @staticmethod
    def mul(a, b):                                     # fields.py:167
        GFp2.M += 1
        return _fp2("FP2_MUL", a, b)

    @staticmethod
    def sqr(a):                                        # fields.py:176
        GFp2.S += 1
        return _fp2("FP2_SQR", a)

    @staticmethod
    def neg(a):                                        # fields.py:184
        GFp2.A += 1
        return _fp2("FP2_NEG", a)

    @staticmethod
    def conj(a):                                       # fields.py:189
        GFp2.A += 0.5
        return _fp2("FP2_CONJ", a)

    @staticmethod
    def inv(a):                                        # fields.py:194-199: I + 1; the norm's inversion counts in GFp (2 sqr, 1 add, the chain);
        GFp2.I += 1                                    # its own mul and conj are taken back there ("to avoid double-counting"): net M, A unchanged
        GFp._count("S", 2 + 126)
        GFp._count("A", 1)
        GFp._count("M", 12)
        return _fp2("FP2_INV", a)

    @staticmethod
    def invsqrt(a):                                    # fields.py:202-230 (unused by the reference; restated as written,
        if a[1] == 0:                                  # including `== -1` tests that can never hold)
            t = GFp.invsqrt(a[0])
            return (t, 0) if GFp.mul(a[0], GFp.sqr(t)) == 1 else (0, t)
        n = GFp.add(GFp.sqr(a[0]), GFp.sqr(a[1]))
        s = GFp.invsqrt(n)
        c = GFp.mul(n, s)
        delta = GFp.mul(GFp.add(a[0], c), GFp.half)
        g = GFp.invsqrt(delta)
        h = GFp.mul(delta, g)
        return (GFp.mul(h, s), GFp.neg(GFp.mul(GFp.mul(GFp.mul(a[1], s), g), GFp.half)))

    @staticmethod
    def select(c, x, y):                               # fields.py:237
        return tuple(_select("FP2_SELECT", c, x, y))
