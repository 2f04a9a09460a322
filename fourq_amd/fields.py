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
    half = 1 << 126                                    # fields.py:16

    @staticmethod
    def add(x, y):                                     # fields.py:30
        return _fp("FP_ADD", x, y)

    @staticmethod
    def sub(x, y):                                     # fields.py:36
        return _fp("FP_SUB", x, y)

    @staticmethod
    def mul(x, y):                                     # fields.py:42
        return _fp("FP_MUL", x, y)

    @staticmethod
    def sqr(x):                                        # fields.py:48
        return _fp("FP_SQR", x)

    @staticmethod
    def neg(x):                                        # fields.py:54
        return _fp("FP_NEG", x)

    @staticmethod
    def inv(x):                                        # fields.py:67
        return _fp("FP_INV", x)

    @staticmethod
    def invsqrt(x):                                    # fields.py:110
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
    zero, one, two = (0, 0), (1, 0), (2, 0)            # fields.py:140-142
    half = (GFp.half, 0)

    @staticmethod
    def add(a, b):                                     # fields.py:157
        return _fp2("FP2_ADD", a, b)

    @staticmethod
    def sub(a, b):                                     # fields.py:162
        return _fp2("FP2_SUB", a, b)

    @staticmethod
    def mul(a, b):                                     # fields.py:167
        return _fp2("FP2_MUL", a, b)

    @staticmethod
    def sqr(a):                                        # fields.py:176
        return _fp2("FP2_SQR", a)

    @staticmethod
    def neg(a):                                        # fields.py:184
        return _fp2("FP2_NEG", a)

    @staticmethod
    def conj(a):                                       # fields.py:189
        return _fp2("FP2_CONJ", a)

    @staticmethod
    def inv(a):                                        # fields.py:194
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
