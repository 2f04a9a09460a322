"""Drop-in mirror of the reference's `impl/curve4q.py` API, computed on the MI355X.

Same names, argument meaning, value conventions (nested tuples of Python ints) and error
behaviour as bifurcation/fourq; every function runs on the GPU through libfourq_amd.so -- there
is no CPU arithmetic here beyond packing tuples into words.  Single calls are batches of one;
use `fourq_amd.Engine` (or the `*_batch` helpers below) for throughput.

    reference symbol (curve4q.py)                 here
    MUL_windowed :188 / MUL_endo :405             same, plus aliases mul_windowed / mul
    table_windowed :179 / table_endo :385         same
    DH_core :446, DH_windowed :464, DH_endo :467  same; dh_exchange = DH_endo(a, DH_endo(b, G))
    PointOnCurve :23, AffineToR1 :100, R1toAffine :103, R1toR2 :109, R1toR3 :119, R2toR4 :129,
    DBL :138, ADD_core :155, ADD :174, tau :258, tau_dual :269, upsilon :282, chi :304,
    phi :318, psi :321, decompose :339, recode :358
"""
import os
import threading

import numpy as np

from . import codec
from .combine import Combiner
from .constants import N, Gx, Gy, Ox, Oy, d, P127  # noqa: F401  (re-exported, as the reference does)
from .engine import default_engine
from .fields import GFp, GFp2  # noqa: F401

p1271 = P127
# the reference's remaining module-level constants under their own names (curve4q.py:240-256, :326-337)
from .constants import ctau, ctaudual  # noqa: E402,F401
from . import constants as _k  # noqa: E402
cphi0, cphi1, cphi2, cphi3, cphi4, cphi5, cphi6, cphi7, cphi8, cphi9 = _k.cphi
cpsi1, cpsi2, cpsi3, cpsi4 = (_k.cpsi[i] for i in (1, 2, 3, 4))
b1, b2, b3, b4 = (list(row) for row in _k.BASIS)
L1, L2, L3, L4 = _k.ELL
c, cp = list(_k.OFFSET_C), list(_k.OFFSET_CP)

_MSG = {1: "Point not on curve", 2: "DH computation resulted in neutral point"}   # curve4q.py:448, :460


def _prim_pt(op, *points):
    row = np.concatenate([codec.pack_point(P) for P in points]).reshape(1, -1)
    return codec.unpack_fp2s(default_engine().prim(op, row)[0])


# ---- point encoding / decoding (curve4q.py:33-96) --------------------------------------------------
_DECODE_EXC = {
    1: (Exception, "Malformed point: reserved bit is not zero"),                      # curve4q.py:53, :62
    2: (Exception, "Point not on curve"),                                             # curve4q.py:94
    3: (AttributeError, "type object 'GFp' has no attribute 'two'"),                  # curve4q.py:77 (reference bug, kept)
}


def sign(X):                                                                          # curve4q.py:33-39
    x0, x1 = X[0] % P127, X[1] % P127
    return (x0 >> 126) if x0 != 0 else (x1 >> 126)


def encode(X, Y):
    """32-byte encoding of the affine point (X, Y) as a bytearray (curve4q.py:41-46)."""
    row = codec.pack_point((X, Y)).reshape(1, 8)
    return bytearray(default_engine().encode(row)[0].tobytes())


def decode(B):
    """Affine point of a 32-byte encoding; raises what the reference raises (curve4q.py:49-96).  Unlike the
    reference the caller's buffer is left untouched (the reference clears the sign bit in place, :56)."""
    if len(B) != 32:
        raise Exception("Malformed point: length {} != 32".format(len(B)))
    out, status = default_engine().decode(np.frombuffer(bytes(B), dtype=np.uint8).reshape(1, 32))
    if status[0]:
        exc, msg = _DECODE_EXC[int(status[0])]
        raise exc(msg)
    return codec.unpack_fp2s(out[0])


# ---- membership and representations ------------------------------------------------------------
def PointOnCurve(P):
    (X, Y) = P
    return bool(default_engine().prim("PT_ON_CURVE", codec.pack_point((X, Y)).reshape(1, -1))[0, 0])


def AffineToR1(X, Y):
    return (X, Y, GFp2.one, X, Y)


def R1toAffine(P):
    (X, Y, Z, Ta, Tb) = P
    return _prim_pt("PT_R1TOAFFINE", P)


def R1toR2(P):
    (X, Y, Z, Ta, Tb) = P
    return _prim_pt("PT_R1TOR2", P)


def R1toR3(P):
    (X, Y, Z, Ta, Tb) = P
    return _prim_pt("PT_R1TOR3", P)


def R2toR4(P):
    (N_, D_, E_, F_) = P
    return _prim_pt("PT_R2TOR4", P)


def DBL(P):
    (X1, Y1, Z1) = P[:3]
    return _prim_pt("PT_DBL", (X1, Y1, Z1, GFp2.zero, GFp2.zero))


def ADD_core(P, Q):
    (N1, D1, E1, F1) = P
    (N2, D2, Z2, T2) = Q
    return _prim_pt("PT_ADD_CORE", P, Q)


def ADD(P, Q):
    (X, Y, Z, Ta, Tb) = P
    (N2, D2, Z2, T2) = Q
    return _prim_pt("PT_ADD", P, Q)


# ---- endomorphisms -----------------------------------------------------------------------------
def tau(P):
    (X1, Y1, Z1) = P
    return _prim_pt("PT_TAU", P)


def tau_dual(P):
    (X1, Y1, Z1) = P
    return _prim_pt("PT_TAU_DUAL", P)


def upsilon(P):
    (X1, Y1, Z1) = P
    return _prim_pt("PT_UPSILON", P)


def chi(P):
    (X1, Y1, Z1) = P
    return _prim_pt("PT_CHI", P)


def phi(P):
    return _prim_pt("PT_PHI", tuple(P[:3]) + (GFp2.zero, GFp2.zero))


def psi(P):
    return _prim_pt("PT_PSI", tuple(P[:3]) + (GFp2.zero, GFp2.zero))


# ---- recoding ----------------------------------------------------------------------------------
def decompose(m):
    out = default_engine().prim("SC_DECOMPOSE", codec.pack_scalars([m]))[0]
    return [int(x) for x in out]


def recode(v):
    """(m, d): 65 sign bits and 65 digits of the decomposed scalar v (curve4q.py:358-380)."""
    (v1, v2, v3, v4) = v
    row = np.array([[v1, v2, v3, v4]], dtype=np.uint64)
    sign, d0, d1, d2, top = (int(x) for x in default_engine().prim("SC_RECODE", row)[0])
    signs = [(sign >> i) & 1 for i in range(64)] + [1]
    digits = [((d0 >> i) & 1) | (((d1 >> i) & 1) << 1) | (((d2 >> i) & 1) << 2) for i in range(64)] + [top]
    return (signs, digits)


def windowed_digits(m):
    """(sgn[0..62], ind[0..62]) of MUL_windowed (curve4q.py:216-226)."""
    raw = default_engine().prim("SC_WINDOWED", codec.pack_scalars([_reduce_windowed(m)]))[0].tobytes()
    return [b >> 3 for b in raw[:63]], [b & 7 for b in raw[:63]]


# ---- tables and scalar multiplication ----------------------------------------------------------
def table_windowed(P):
    (X, Y, Z, Ta, Tb) = P
    return codec.unpack_table(default_engine().table_windowed(codec.pack_point(P)))


def table_endo(P):
    (X, Y, Z, Ta, Tb) = P
    return codec.unpack_table(default_engine().table_endo(codec.pack_point(P)))


def _reduce_windowed(m):
    # MUL_windowed starts with `reduced = m % N` (curve4q.py:217), so any integer is acceptable
    return m if 0 <= m < (1 << 256) else m % N


def _check_endo_scalar(m):
    if not 0 <= m < (1 << 256):
        raise ValueError("MUL_endo takes a scalar in [0, 2^256) (it is used unreduced, curve4q.py:433)")
    return m


# Single variable-base calls made from several threads at once go to the GPU as ONE batch (fourq_amd/combine.py): the calls that
# arrive while a batch is running are taken along by the next one.  A lone call is not delayed.  FOURQ_COMBINE=0 turns it off.
_COMBINE = os.environ.get("FOURQ_COMBINE", "1") != "0"
_combiners = {}


def _combined(kind, *rows):
    """rows: this call's packed arguments, each (1, words).  Returns this call's (output row, status or None)."""
    cb = _combiners.get(kind)
    if cb is None:
        cb = _combiners.setdefault(kind, Combiner(lambda items, kind=kind: _run_batch(kind, items)))
    return cb(*rows)


# DH_*(m, G) on the curve's own generator -- key generation, the commonest DH call -- goes through the fixed-base comb of [392]G
# (fourq_comb_mul_batch: 6 doublings + 27 mixed additions instead of table construction + 64 steps; the draft allows any method that
# agrees on all inputs, draft-ladd-cfrg-4q.md:725-729; outputs are affine, hence identical).  FOURQ_COMB_KEYGEN=0 turns it off.
_COMB_KEYGEN = os.environ.get("FOURQ_COMB_KEYGEN", "1") != "0"
_g_comb_table = None
_g_comb_lock = threading.Lock()


def _g_comb():
    global _g_comb_table
    if _g_comb_table is None:
        with _g_comb_lock:
            if _g_comb_table is None:
                eng = default_engine()
                g392 = eng.mul_endo(codec.pack_scalars([392]), codec.pack_point((Gx, Gy, (1, 0), Gx, Gy)).reshape(1, 20))[0]
                _g_comb_table = eng.comb_table(g392)
    return _g_comb_table


def _run_batch(kind, items):
    eng = default_engine()
    cols = [np.concatenate(c) if len(items) > 1 else c[0] for c in zip(*items)]
    if kind[0] == "keygen":
        out, status = eng.comb_mul(cols[0], _g_comb())
        return [(out[i], int(status[i])) for i in range(len(items))]
    if kind[0] == "mul":
        out = (eng.mul_endo if kind[1] == "endo" else eng.mul_windowed)(*cols)
        return [(out[i], None) for i in range(len(items))]
    out, status = (eng.dh_endo if kind[1] == "endo" else eng.dh_windowed)(*cols)
    return [(out[i], int(status[i])) for i in range(len(items))]


def combine_stats():
    """{kind: {"calls", "batches", "largest_batch"}} of the combined single calls so far."""
    return {"%s_%s" % k: cb.stats() for k, cb in list(_combiners.items())}


def _mul(kind, s, P, table):
    eng = default_engine()
    if table:                                              # `if not T` in the reference, curve4q.py:211
        out = (eng.mul_endo_fixed if kind == "endo" else eng.mul_windowed_fixed)(s, codec.pack_table(table))[0]
    elif _COMBINE:
        out, _ = _combined(("mul", kind), s, codec.pack_point(P).reshape(1, 20))
    else:
        out = (eng.mul_endo if kind == "endo" else eng.mul_windowed)(s, codec.pack_point(P).reshape(1, 20))[0]
    return codec.unpack_fp2s(out)


def MUL_windowed(m, P, table=None):
    (X, Y, Z, Ta, Tb) = P                                  # shape check, curve4q.py:190
    return _mul("windowed", codec.pack_scalars([_reduce_windowed(m)]), P, table)


def MUL_endo(m, P, table=None):
    (X, Y, Z, Ta, Tb) = P                                  # shape check, curve4q.py:407
    return _mul("endo", codec.pack_scalars([_check_endo_scalar(m)]), P, table)


def MUL_windowed_batch(ms, Ps=None, table=None):
    """[MUL_windowed(m_i, P_i)] (table=None) or [MUL_windowed(m_i, ., table)] in one launch."""
    s = codec.pack_scalars([_reduce_windowed(m) for m in ms])
    eng = default_engine()
    out = eng.mul_windowed_fixed(s, codec.pack_table(table)) if table else eng.mul_windowed(s, codec.pack_points(Ps, 5))
    return codec.unpack_points(out)


def MUL_endo_batch(ms, Ps=None, table=None):
    s = codec.pack_scalars([_check_endo_scalar(m) for m in ms])
    eng = default_engine()
    out = eng.mul_endo_fixed(s, codec.pack_table(table)) if table else eng.mul_endo(s, codec.pack_points(Ps, 5))
    return codec.unpack_points(out)


# ---- Diffie-Hellman ----------------------------------------------------------------------------
def _dh(kind, m, P, table):
    (X, Y) = P
    eng = default_engine()
    s = codec.pack_scalars([_reduce_windowed(m) if kind == "windowed" else _check_endo_scalar(m)])
    if _COMB_KEYGEN and not table and (X, Y) == (Gx, Gy):
        if _COMBINE:
            out, status = _combined(("keygen", "comb"), s)
        else:
            out, status = eng.comb_mul(s, _g_comb())
            out, status = out[0], int(status[0])
        if status:
            raise Exception(_MSG[status])
        return codec.unpack_fp2s(out)
    pts = codec.pack_point((X, Y)).reshape(1, 8)
    if table or not _COMBINE:
        out, status = (eng.dh_windowed if kind == "windowed" else eng.dh_endo)(s, pts, codec.pack_table(table) if table else None)
        out, status = out[0], int(status[0])
    else:
        out, status = _combined(("dh", kind), s, pts)
    if status:
        raise Exception(_MSG[status])
    return codec.unpack_fp2s(out)


def DH_core(m, P, mul, table=None):
    """curve4q.py:446-462.  The two multiplications of the reference run as one fused DH kernel; any other callable
    `mul(m, Q, table=...)` gets the reference's own composition, each step on the GPU through the primitives."""
    if mul is MUL_windowed:
        return _dh("windowed", m, P, table)
    if mul is MUL_endo:
        return _dh("endo", m, P, table)
    if not PointOnCurve(P):
        raise Exception(_MSG[1])
    Q = _prim_pt("PT_COFACTOR392", (P[0], P[1]))           # the DBL/ADD chain of curve4q.py:450-455
    Q = R1toAffine(mul(m, Q, table=table))
    if Q == (Ox, Oy):
        raise Exception(_MSG[2])
    return Q


def DH_windowed(m, P, table=None):
    return _dh("windowed", m, P, table)


def DH_endo(m, P, table=None):
    return _dh("endo", m, P, table)


def DH_batch(kind, ms, Ps, table=None):
    """Batched DH_<kind>: returns (list of affine points or None, list of exception messages or None)."""
    eng = default_engine()
    red = _reduce_windowed if kind == "windowed" else _check_endo_scalar
    s = codec.pack_scalars([red(m) for m in ms])
    t = codec.pack_table(table) if table else None
    out, status = (eng.dh_windowed if kind == "windowed" else eng.dh_endo)(s, codec.pack_points(Ps, 2), t)
    pts = codec.unpack_points(out)
    return [p if st == 0 else None for p, st in zip(pts, status)], [_MSG.get(int(st)) for st in status]


# BASELINE.json spellings
mul = MUL_endo
mul_windowed = MUL_windowed


def dh_exchange(a, b, table=None):
    """DH_endo(a, DH_endo(b, G)); `table` = table_endo([392]G) accelerates the fixed-base half."""
    return DH_endo(a, DH_endo(b, (Gx, Gy), table=table))
