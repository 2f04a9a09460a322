"""Pins the CPU oracle (oracle/curve4q_oracle.py) against the reference's own vectors.

kat.json holds the literal known answers of the reference self-tests (curve4q.py:473-778,
fields.py:366-403); the other files are outputs of the reference itself run in the build
container (tests/golden/make_golden.py).  No GPU needed.
"""
import random

import pytest

import curve4q_oracle as o
from conftest import unhex


def affine(P):
    zi = o.f2_inv(P[2])
    return (o.f2_mul(P[0], zi), o.f2_mul(P[1], zi))


G1 = o.AffineToR1(o.Gx, o.Gy)


def test_constants(golden):
    kat = golden("kat.json", raw=True)
    assert o.P127 == int(kat["p1271"], 16)
    assert o.d == unhex(kat["d"]) and o.N == int(kat["N"], 16)
    assert (o.Gx, o.Gy) == unhex(kat["G"]) and (o.Ox, o.Oy) == unhex(kat["O"])
    assert o.PointOnCurve((o.Ox, o.Oy)) and o.PointOnCurve((o.Gx, o.Gy))  # curve4q.py:473-475


def test_field_literals(golden):  # fields.py:366-399
    kat = golden("kat.json", raw=True)
    assert o.fp_mul(o.fp_inv(13), 13) == 1
    assert o.fp_mul(13, o.fp_sqr(o.fp_invsqrt(13))) == 1
    for row in kat["gfp2_literals"]:
        op, args, want = row[0], [unhex(a) for a in row[1:-1]], unhex(row[-1])
        assert getattr(o.GFp2, op)(*args) == want
    assert o.f2_mul((2, 3), o.f2_inv((2, 3))) == (1, 0)
    assert o.f2_select(1, (2, 3), (5, 7)) == (2, 3) and o.f2_select(0, (2, 3), (5, 7)) == (5, 7)


def test_reps(golden):  # curve4q.py:490-511
    r = unhex(golden("kat.json", raw=True)["reps"])
    assert o.R1toR2(r["r1"]) == r["r2"]
    assert o.R1toR3(r["r1"]) == r["r3"]
    assert o.R2toR4(r["r2"]) == r["r4"]


def test_core_kats(golden):  # curve4q.py:513-547
    kat = golden("kat.json", raw=True)
    doubleP, P1000 = unhex(kat["doubleP"]), unhex(kat["P1000"])
    A = (o.Gx, o.Gy, o.F2_ONE)
    for _ in range(1000):
        A = o.DBL(A)[:3]
    assert affine(A) == doubleP
    O1 = o.AffineToR1(o.Ox, o.Oy)
    assert affine(o.ADD(G1, o.R1toR2(O1))) == (o.Gx, o.Gy)
    assert affine(o.ADD(O1, o.R1toR2(G1))) == (o.Gx, o.Gy)
    Pt = G1
    for _ in range(1000):
        Pt = o.ADD(Pt, o.R1toR2(Pt))
    assert affine(Pt) == doubleP
    Q = o.R1toR2(G1)
    Pt = o.DBL(G1[:3])
    for _ in range(1000):
        Pt = o.ADD(Pt, Q)
    assert affine(Pt) == P1000


def test_endo_kats(golden):  # curve4q.py:600-617
    kat = golden("kat.json", raw=True)
    Pt = Qt = G1
    for _ in range(1000):
        Pt, Qt = o.phi(Pt), o.psi(Qt)
    assert affine(Pt) == unhex(kat["phiP"])
    assert affine(Qt) == unhex(kat["psiP"])


def test_decompose_kats(golden):  # curve4q.py:623-638
    for m, v in unhex(golden("kat.json", raw=True)["decompose"]):
        assert tuple(o.decompose(m)) == v


@pytest.mark.parametrize("mul", [o.MUL_windowed, o.MUL_endo], ids=["windowed", "endo"])
def test_mul_chain_kat(golden, mul):  # curve4q.py:549-567, :579, :685
    kat = golden("kat.json", raw=True)
    assert list(unhex(kat["mul_chain_seed"])) == [0x3AD457AB55456230, 0x3A8B3C2C6FD86E0C,
                                                  0x7E38F7C9CFBB9166, 0x0028FD6CBDA458F0]
    A = G1
    for m in o.kat_scalars(1000):
        A = mul(m, A)
    assert affine(A) == unhex(kat["mulP"])


def test_mul_small_and_fixed():  # curve4q.py:571-598, :677-704
    rng = random.Random(7)
    for mul, table in ((o.MUL_windowed, o.table_windowed), (o.MUL_endo, o.table_endo)):
        T = table(G1)
        assert affine(mul(1, G1)) == (o.Gx, o.Gy) == affine(mul(1, G1, table=T))
        assert affine(mul(2, G1)) == affine(o.DBL(G1)) == affine(mul(2, G1, table=T))
        for _ in range(5):
            m = rng.getrandbits(256)
            assert mul(m, G1, table=T) == mul(m, G1)


def test_recode_reconstructs():  # curve4q.py:640-673
    rng = random.Random(8)
    for _ in range(200):
        v = o.decompose(rng.getrandbits(256))
        s, dg = o.recode(v)
        a = [0, 0, 0, 0]
        for i in range(64, -1, -1):
            sg = 1 if s[i] else -1
            a = [2 * a[0] + sg] + [2 * a[j] + sg * ((dg[i] >> (j - 1)) & 1) for j in (1, 2, 3)]
        assert a == list(v)


def test_encode(golden):  # curve4q.py:478-481
    assert bytes(o.encode(o.Gx, o.Gy)).hex() == golden("kat.json", raw=True)["Genc"]


def test_dh_properties(golden):  # curve4q.py:706-778
    rng = random.Random(9)
    G = (o.Gx, o.Gy)
    for dh in (o.DH_windowed, o.DH_endo):
        Pt = G
        for _ in range(3):
            m = rng.getrandbits(256)
            Q1 = dh(m, Pt)
            assert Q1 == o.R1toAffine(o.MUL_windowed(392 * m, o.AffineToR1(*Pt)))
            Pt = Q1
        a, b = rng.getrandbits(256), rng.getrandbits(256)
        assert dh(a, dh(b, G)) == dh(b, dh(a, G))
    with pytest.raises(Exception, match="Point not on curve"):
        o.DH_endo(1, ((0, 0), (0, 0)))
    with pytest.raises(Exception, match="neutral point"):
        o.DH_endo(1, unhex(golden("kat.json", raw=True)["P392"]))


# ----------------------------------------------------------- reference-generated vectors
def test_field_vectors(golden):
    g = golden("field.json")
    for a, b, add, sub, mul, sqr, neg in g["fp"]:
        assert (o.fp_add(a, b), o.fp_sub(a, b), o.fp_mul(a, b), o.fp_sqr(a), o.fp_neg(a)) == (add, sub, mul, sqr, neg)
    for a, b, add, sub, mul, sqr, neg, conj in g["fp2"]:
        assert (o.f2_add(a, b), o.f2_sub(a, b), o.f2_mul(a, b)) == (add, sub, mul)
        assert (o.f2_sqr(a), o.f2_neg(a), o.f2_conj(a)) == (sqr, neg, conj)
    for a, inv, isq in g["fp_inv"]:
        assert (o.fp_inv(a), o.fp_invsqrt(a)) == (inv, isq)
    for a, inv in g["fp2_inv"]:
        assert o.f2_inv(a) == inv


def test_group_vectors(golden):
    g = golden("group.json")
    for Pt, dbl, r2, r3, r4 in g["r1"]:
        assert (o.DBL(Pt), o.R1toR2(Pt), o.R1toR3(Pt), o.R2toR4(o.R1toR2(Pt))) == (dbl, r2, r3, r4)
    for Pt, Q2, add, addc in g["add"]:
        assert o.ADD(Pt, Q2) == add and o.ADD_core(o.R1toR3(Pt), Q2) == addc
    for Pt, t, td, up, ch, ph, ps in g["endo"]:
        assert o.tau(Pt[:3]) == t
        assert (o.tau_dual(t), o.upsilon(t), o.chi(t), o.phi(Pt), o.psi(Pt)) == (td, up, ch, ph, ps)
    for a, ok in g["on_curve"]:
        assert o.PointOnCurve(a) == ok


def test_recode_vectors(golden):
    g = golden("recode.json", raw=True)
    for m, v in g["decompose"]:
        assert o.decompose(int(m, 16)) == [int(x, 16) for x in v]
    for m, s, dg in g["recode"]:
        ss, dd = o.recode(o.decompose(int(m, 16)))
        assert "".join(map(str, ss)) == s and "".join(map(str, dd)) == dg
    for m, digits in g["windowed"]:
        sgn, ind = o.recode_windowed(int(m, 16))
        assert sgn == [1 if x > 0 else 0 for x in digits]
        assert ind == [(abs(x) - 1) // 2 for x in digits]
        assert digits[62] == 1


def test_table_vectors(golden):
    for Pt, tw, te in golden("tables.json")["tables"]:
        assert tuple(o.table_windowed(Pt)) == tw and tuple(o.table_endo(Pt)) == te


def test_mul_vectors(golden):
    g = golden("mul.json")
    for m, Pt, e, w in g["var"] + g["edge"]:
        assert o.MUL_endo(m, Pt) == e and o.MUL_windowed(m, Pt) == w
    for blk in g["fixed"]:
        for m, e, w in blk["rows"]:
            assert o.MUL_endo(m, blk["P"], table=list(blk["table_endo"])) == e
            assert o.MUL_windowed(m, blk["P"], table=list(blk["table_windowed"])) == w


def test_dh_vectors(golden):
    g = golden("dh.json", raw=True)
    for m, Pt, e, w in unhex(g["dh"]):
        assert o.DH_endo(m, Pt) == e and o.DH_windowed(m, Pt) == w
    fx = g["fixed"]
    te, tw = list(unhex(fx["table_endo"])), list(unhex(fx["table_windowed"]))
    for m, e, w in unhex(fx["rows"]):
        assert o.DH_endo(m, (o.Gx, o.Gy), table=te) == e
        assert o.DH_windowed(m, (o.Gx, o.Gy), table=tw) == w
    for m, Pt, msg in g["reject"]:
        for dh in (o.DH_endo, o.DH_windowed):
            with pytest.raises(Exception) as ei:
                dh(int(m, 16), unhex(Pt))
            assert str(ei.value) == msg
    for a, b, ab in unhex(g["exchange"]):
        assert o.dh_exchange(a, b) == ab


def test_wire_vectors(golden):
    """encode / decode (curve4q.py:33-96) incl. the exceptions the reference raises, by type and message."""
    w = golden("wire.json", raw=True)
    for Pt, enc in w["roundtrip"]:
        Pt = unhex(Pt)
        assert bytes(o.encode(Pt[0], Pt[1])).hex() == enc and o.decode(bytes.fromhex(enc)) == Pt
    for row in w["strings"] + w["malformed"]:
        raw = bytes.fromhex(row[0])
        if row[1] == "ok":
            assert o.decode(raw) == unhex(row[2])
        else:
            with pytest.raises(Exception) as ei:
                o.decode(raw)
            assert (type(ei.value).__name__, str(ei.value)) == (row[1], row[2])


def test_protocol_vectors(golden):
    """protocol.json: the reference's GFp.select / GFp2.select and its own composition encode(DH(m, decode(key))), every
    failure included (draft-ladd-cfrg-4q.md:707-723)."""
    g = golden("protocol.json", raw=True)
    for c, x, y, r in unhex(g["select"]):
        assert o.fp_select(c, x, y) == r
    for c, a, b, r in unhex(g["select2"]):
        assert o.f2_select(c, a, b) == r
    for m, key, endo, win in g["dh_bytes"]:
        for dh, want in ((o.DH_endo, endo), (o.DH_windowed, win)):
            try:
                got = ["ok", bytes(o.encode(*dh(int(m, 16), o.decode(bytes.fromhex(key))))).hex()]
            except Exception as exc:
                got = [type(exc).__name__, str(exc)]
            assert got == want, (m, key)
