"""GPU parity, layer by layer: every device function against the reference-generated golden
vectors and against the Python oracle on seeded random + edge inputs.  Bit-exact (integers)."""
import random

import numpy as np
import pytest

import curve4q_oracle as o
from fourq_amd import codec

pytestmark = pytest.mark.gpu

P = o.P127


@pytest.fixture(scope="module")
def eng():
    from fourq_amd import Engine
    e = Engine(0)
    yield e
    e.close()


def fp_rows(pairs):
    return np.array([[a & codec.M64, a >> 64, b & codec.M64, b >> 64] for a, b in pairs], dtype=np.uint64)


def fp_out(arr):
    return [int(r[0]) | (int(r[1]) << 64) for r in arr]


def f2_rows(pairs):
    return np.array([[w for e in (a, b) for c in e for w in (c & codec.M64, c >> 64)] for a, b in pairs], dtype=np.uint64)


def f2_out(arr):
    return [codec.unpack_fp2s(r)[0] for r in arr]


def test_fp_golden(eng, golden):
    rows = golden("field.json")["fp"]
    x = fp_rows([(r[0], r[1]) for r in rows])
    for k, op in enumerate(["FP_ADD", "FP_SUB", "FP_MUL", "FP_SQR", "FP_NEG"]):
        assert fp_out(eng.prim(op, x)) == [r[2 + k] for r in rows], op
    inv = golden("field.json")["fp_inv"]
    assert fp_out(eng.prim("FP_INV", fp_rows([(r[0], 0) for r in inv]))) == [r[1] for r in inv]


def test_fp2_golden(eng, golden):
    rows = golden("field.json")["fp2"]
    x = f2_rows([(r[0], r[1]) for r in rows])
    for k, op in enumerate(["FP2_ADD", "FP2_SUB", "FP2_MUL", "FP2_SQR", "FP2_NEG", "FP2_CONJ"]):
        assert f2_out(eng.prim(op, x)) == [r[2 + k] for r in rows], op
    inv = golden("field.json")["fp2_inv"]
    assert f2_out(eng.prim("FP2_INV", f2_rows([(r[0], (0, 0)) for r in inv]))) == [r[1] for r in inv]


def test_field_random_and_noncanonical(eng):
    """8192 seeded pairs incl. values the 128-bit container can hold but that are >= p."""
    rng = random.Random(31337)
    edge = [0, 1, 2, P - 1, P, P + 1, (1 << 127), (1 << 128) - 1, (1 << 64) - 1, 1 << 64, (1 << 126), (1 << 104) - 1,
            (1 << 78), (1 << 52) - 1, (1 << 26), (1 << 26) - 1]
    vals = edge + [rng.getrandbits(128) for _ in range(96)]
    pairs = [(a, b) for a in edge for b in edge] + [(rng.choice(vals), rng.choice(vals)) for _ in range(4096)]
    x = fp_rows(pairs)
    for op, fn in (("FP_ADD", o.fp_add), ("FP_SUB", o.fp_sub), ("FP_MUL", o.fp_mul)):
        assert fp_out(eng.prim(op, x)) == [fn(a % P, b % P) for a, b in pairs], op
    assert fp_out(eng.prim("FP_SQR", x)) == [o.fp_sqr(a % P) for a, _ in pairs]
    assert fp_out(eng.prim("FP_NEG", x)) == [o.fp_neg(a % P) for a, _ in pairs]
    f2 = [((rng.choice(vals), rng.choice(vals)), (rng.choice(vals), rng.choice(vals))) for _ in range(4096)]
    red = [(tuple(c % P for c in a), tuple(c % P for c in b)) for a, b in f2]
    y = f2_rows(f2)
    for op, fn in (("FP2_ADD", o.f2_add), ("FP2_SUB", o.f2_sub), ("FP2_MUL", o.f2_mul)):
        assert f2_out(eng.prim(op, y)) == [fn(a, b) for a, b in red], op
    assert f2_out(eng.prim("FP2_SQR", y)) == [o.f2_sqr(a) for a, _ in red]
    assert f2_out(eng.prim("FP2_INV", y[:64])) == [o.f2_inv(a) for a, _ in red[:64]]


def test_group_golden(eng, golden):
    g = golden("group.json")
    pts = codec.pack_points([r[0] for r in g["r1"]], 5)
    assert codec.unpack_points(eng.prim("PT_DBL", pts)) == [r[1] for r in g["r1"]]
    assert codec.unpack_points(eng.prim("PT_R1TOR2", pts)) == [r[2] for r in g["r1"]]
    assert codec.unpack_points(eng.prim("PT_R1TOR3", pts)) == [r[3] for r in g["r1"]]
    r2 = codec.pack_points([r[2] for r in g["r1"]], 4)
    assert codec.unpack_points(eng.prim("PT_R2TOR4", r2)) == [r[4] for r in g["r1"]]
    both = np.concatenate([codec.pack_points([r[0] for r in g["add"]], 5), codec.pack_points([r[1] for r in g["add"]], 4)], axis=1)
    assert codec.unpack_points(eng.prim("PT_ADD", both)) == [r[2] for r in g["add"]]
    r3 = [o.R1toR3(r[0]) for r in g["add"]]
    both = np.concatenate([codec.pack_points(r3, 4), codec.pack_points([r[1] for r in g["add"]], 4)], axis=1)
    assert codec.unpack_points(eng.prim("PT_ADD_CORE", both)) == [r[3] for r in g["add"]]


def test_endomorphism_golden(eng, golden):
    rows = golden("group.json")["endo"]
    p3 = codec.pack_points([r[0][:3] for r in rows], 3)
    t = codec.pack_points([r[1] for r in rows], 3)
    assert codec.unpack_points(eng.prim("PT_TAU", p3)) == [r[1] for r in rows]
    assert codec.unpack_points(eng.prim("PT_TAU_DUAL", t)) == [r[2] for r in rows]
    assert codec.unpack_points(eng.prim("PT_UPSILON", t)) == [r[3] for r in rows]
    assert codec.unpack_points(eng.prim("PT_CHI", t)) == [r[4] for r in rows]
    p5 = codec.pack_points([r[0] for r in rows], 5)
    assert codec.unpack_points(eng.prim("PT_PHI", p5)) == [r[5] for r in rows]
    assert codec.unpack_points(eng.prim("PT_PSI", p5)) == [r[6] for r in rows]


def test_on_curve_cofactor_affine(eng, golden):
    rows = golden("group.json")["on_curve"]
    got = eng.prim("PT_ON_CURVE", codec.pack_points([r[0] for r in rows], 2))[:, 0]
    assert [bool(v) for v in got] == [r[1] for r in rows]
    good = [r[0] for r in rows if r[1]]
    got = codec.unpack_points(eng.prim("PT_COFACTOR392", codec.pack_points(good, 2)))
    assert got == [o.clear_cofactor(o.AffineToR1(*a)) for a in good]
    r1 = [r[0] for r in golden("group.json")["r1"][:12] if r[0][2] != (0, 0)]
    got = codec.unpack_points(eng.prim("PT_R1TOAFFINE", codec.pack_points(r1, 5)))
    assert got == [o.R1toAffine(p) for p in r1]


def test_kat_iterated_primitives(eng, golden):
    """1000 x DBL, 1000 x phi, 1000 x psi chains of the reference self-tests (curve4q.py:517-522, :603-617),
    run as 1000 dependent single-element launches through the C ABI."""
    from conftest import unhex
    kat = golden("kat.json", raw=True)
    G1 = codec.pack_point(o.AffineToR1(o.Gx, o.Gy)).reshape(1, 20)
    for op, key in (("PT_DBL", "doubleP"), ("PT_PHI", "phiP"), ("PT_PSI", "psiP")):
        A = G1
        for _ in range(1000):
            A = eng.prim(op, A)
        assert o.R1toAffine(codec.unpack_fp2s(A[0])) == unhex(kat[key]), key


def test_recode_golden(eng, golden):
    g = golden("recode.json", raw=True)
    ms = [int(r[0], 16) for r in g["decompose"]]
    got = eng.prim("SC_DECOMPOSE", codec.pack_scalars(ms))
    assert [[int(x) for x in row] for row in got] == [[int(x, 16) for x in r[1]] for r in g["decompose"]]
    ms = [int(r[0], 16) for r in g["recode"]]
    rec = eng.prim("SC_RECODE", eng.prim("SC_DECOMPOSE", codec.pack_scalars(ms)))
    for row, (_, signs, digits) in zip(rec, g["recode"]):
        sign, d0, d1, d2, top = (int(x) for x in row)
        s = "".join(str((sign >> i) & 1) for i in range(64)) + "1"
        dg = "".join(str(((d0 >> i) & 1) | (((d1 >> i) & 1) << 1) | (((d2 >> i) & 1) << 2)) for i in range(64)) + str(top)
        assert (s, dg) == (signs, digits)
    ms = [int(r[0], 16) for r in g["windowed"]]
    win = eng.prim("SC_WINDOWED", codec.pack_scalars(ms))
    for row, (_, digits) in zip(win, g["windowed"]):
        raw = row.tobytes()[:63]
        assert [b >> 3 for b in raw] == [1 if dgt > 0 else 0 for dgt in digits]
        assert [b & 7 for b in raw] == [(abs(dgt) - 1) // 2 for dgt in digits]


def test_recode_random(eng):
    rng = random.Random(99)
    ms = [rng.getrandbits(256) for _ in range(2048)] + [0, 1, o.N - 1, o.N, o.N + 1, (1 << 256) - 1, 1 << 255]
    got = eng.prim("SC_DECOMPOSE", codec.pack_scalars(ms))
    assert [[int(x) for x in row] for row in got] == [o.decompose(m) for m in ms]
    win = eng.prim("SC_WINDOWED", codec.pack_scalars(ms))
    for row, m in zip(win, ms):
        sgn, ind = o.recode_windowed(m)
        raw = row.tobytes()[:63]
        assert [b >> 3 for b in raw] == sgn and [b & 7 for b in raw] == ind


def test_reference_shaped_field_namespaces():
    """fourq_amd.fields.GFp / GFp2 (the mirror a module swap lands on) against the oracle, one call per function,
    GFp2.invsqrt (fields.py:202-230, composed on the host from GFp primitives) included."""
    import random
    from fourq_amd.fields import GFp, GFp2
    rng = random.Random(99)
    p = o.P127 if hasattr(o, "P127") else (1 << 127) - 1
    for _ in range(6):
        x, y = rng.getrandbits(127) % p, rng.getrandbits(127) % p
        a, b = (x, y), (rng.getrandbits(127) % p, rng.getrandbits(127) % p)
        for name in ("add", "sub", "mul"):
            assert getattr(GFp, name)(x, y) == getattr(o.GFp, name)(x, y)
            assert getattr(GFp2, name)(a, b) == getattr(o.GFp2, name)(a, b)
        for name in ("sqr", "neg", "inv", "invsqrt"):
            assert getattr(GFp, name)(x) == getattr(o.GFp, name)(x)
        for name in ("sqr", "neg", "conj", "inv", "invsqrt"):
            assert getattr(GFp2, name)(a) == getattr(o.GFp2, name)(a), name
        assert GFp2.invsqrt((x, 0)) == o.GFp2.invsqrt((x, 0))
        assert GFp.select(1, x, y) == x and GFp2.select(0, a, b) == b
    assert (GFp2.zero, GFp2.one, GFp2.two, GFp.half) == ((0, 0), (1, 0), (2, 0), 1 << 126)


@pytest.mark.gpu
def test_field_operation_counters_of_the_reference():
    """GFp / GFp2 carry the reference's counters A, S, M, I with ctr() / ctr_reset() (fields.py:10-27, :135-154): every method counts as
    the reference's does (conj half an addition, GFp2.inv one I and nothing else in GFp2, its norm's chain in GFp), and GFp.ctr()
    raises the reference's own NameError (it names a class `GFp1271` that does not exist)."""
    from fourq_amd.fields import GFp, GFp2
    a, b = (3, 5), (7, 11)
    GFp2.ctr_reset(); GFp.ctr_reset()
    GFp2.mul(GFp2.add(a, b), GFp2.sub(a, b)); GFp2.sqr(a); GFp2.neg(b); GFp2.conj(a)
    assert GFp2.ctr() == (3.5, 1, 1, 0)
    assert GFp2.mul(GFp2.inv(a), a) == (1, 0)
    assert GFp2.ctr() == (3.5, 1, 2, 1) and (GFp.A, GFp.S, GFp.M, GFp.I) == (1, 128, 12, 0)
    GFp.ctr_enabled = False
    try:
        assert GFp.mul(GFp.inv(9), 9) == 1 and (GFp.A, GFp.S, GFp.M) == (1, 128, 12)
    finally:
        GFp.ctr_enabled = True
    GFp.add(1, 2); GFp.sub(1, 2); GFp.neg(1); GFp.sqr(3); GFp.mul(3, 4)
    assert (GFp.A, GFp.S, GFp.M) == (4, 129, 13)
    with pytest.raises(NameError, match="GFp1271"):
        GFp.ctr()
    GFp2.ctr_reset(); GFp.ctr_reset()
    assert GFp2.ctr() == (0, 0, 0, 0)
