"""The reference's self-test sequence (impl/curve4q.py:473-790, impl/fields.py:366-409), run against the GPU
engine through the drop-in API.  Prints the same `[PASS] label` / `[FAIL] label` lines as the reference's
`python curve4q.py` and returns the number of failures.

    python -m fourq_amd.selftest            # needs an MI355X

Every value is computed on the device; the expected constants are the FourQlib-derived literals the reference's
tests carry (reproduced here as data) plus the algebraic properties those tests check.
"""
import random
import sys

from . import curve4q as c
from .fields import GFp, GFp2, p1271

KAT = {
    "doubleP": ((0x2C3FD8822C82270FC9099C54855859D6, 0x4DA5B9E83AA7A1B2A7B3F6E2043E8E68),
                (0x2001EB3A576883963EE089F0EB49AA14, 0x0FFDB0D761421F501FEE5617A7E954CD)),      # curve4q.py:520
    "P1000": ((0x3E243958590C4D906480B1EF0A151DB0, 0x5327AF7D84238CD0AA270F644A65D473),
              (0x3EF69A49CB7E02375E06003D73C43EB1, 0x293EB1E26DD23B4E4E752648AC2EF0AB)),        # :545
    "mulP": ((0x257C122BBFC94A1BDFD2B477BD494BEF, 0x469BF80CB5B11F01769593547237C459),
             (0x0901B3817C0E936C281C5067996F3344, 0x570B948EACACE2104FE8C429915F1245)),         # :565
    "phiP": ((0x5550AAB9E7A620EED5B5A3061287DB16, 0x3E61EBB9A1CB0210EC321E6CF33610FC),
             (0x5474BF8EC55603AE7E2851D5A8E83FB9, 0x5476093DBF8BF6BFA5077613491788D5)),         # :607
    "psiP": ((0x75AF54EDB41A2B93D8F3C8C24A2BC7E2, 0x065249F9EDE0C7984DE2466701F009A9),
             (0x06DBB85BFFB7C21E1C6E119ADD608104, 0x060A30903424BF13FD234D6C4CFA3EC1)),         # :615
    "Genc": "87b2cb2b46a224b95a7820a19bee3f0e5c8b4c8444c3a74942020e63f84a1c6e",                  # :478
    "P392": ((0x1318020702DE23BC3C9B73C751B4B192, 0x77AB39A7D8990C0A18E3C409FBD81A95),
             (0x515854B6D19CC2DA1EA2B43B5121A22E, 0x763F89E129497361D74DFF5063E66682)),         # :772
    "decompose": [                                                                               # :623-634
        (0x92990788D66BF558052D112F5498111747B3E28C55984D43FED8C8822AD9F1A7,
         [0xA8EA3F673F711E51, 0xA08D1EAE0B9E071D, 0x55C8DF690050276F, 0x6396739DDA88830F]),
        (0x48E5CA2A675AB49CA214B884813935024B0C61EDC8D1305FE5230DF341623348,
         [0xA53EC4631945B875, 0x521C0BA1261C1934, 0x5C50CE912909185C, 0x93B3C70960B44BAD]),
        (0xAE20E251C36CFA5BE4D9F3D5A5EDFED305A1E8F7F6394D9BE58A15C4B0F1C5E9,
         [0xA621ADA9B3499C9F, 0x7CD17E0095E7AAE6, 0x6E8D23B5BD10BB43, 0x7F18C69F3025234C]),
        (0xB2C950ABC87A55442CC00F1E3AC38F81B7E95036FD191EA134FF616D9806E10C,
         [0x9B30A872EBEA83AF, 0x8F6C73350447C9C3, 0x72FDC76E3456D087, 0x6BA39BA159B0C13D]),
        (0x8E2958A1475ED70762340E9797788E0061F21FCEBD67889FDD4F4CE2B5F6B2DE,
         [0xBE8F3583A0934333, 0xAB45BF6D1BF80B37, 0x4A19FC5CFFE97809, 0x5EA3BAF1A1206442]),
    ],
}


class Report:
    def __init__(self, out=sys.stdout):
        self.out, self.failed = out, 0

    def check(self, label, ok, shown=None):                       # test.test, impl/test.py:21-25
        self.failed += 0 if ok else 1
        self.out.write("[PASS] %s\n" % label if ok else "[FAIL] %s %s\n" % (label, shown if shown is not None else ""))

    def point(self, label, sample, ref):                          # test.testpt, impl/test.py:27-33
        a, b = to_affine(sample), to_affine(ref)
        self.check(label, a == b, a)


def to_affine(P):                                                 # test.toAffine, impl/test.py:11-19
    if len(P) == 2:
        return tuple(P)
    if len(P) not in (3, 5):
        raise Exception("Representation unsupported for normalization")
    zi = GFp2.inv(P[2])
    return (GFp2.mul(P[0], zi), GFp2.mul(P[1], zi))


def kat_scalars(count):                                           # the sequence of curve4q.py:552-559
    s = [0x3AD457AB55456230, 0x3A8B3C2C6FD86E0C, 0x7E38F7C9CFBB9166, 0x0028FD6CBDA458F0]
    for _ in range(count):
        s[1] = s[2]
        s[2] = (s[2] + s[0]) & 0xFFFFFFFFFFFFFFFF
        yield s[0] | (s[1] << 64) | (s[2] << 128) | (s[3] << 192)


def run(loops=1000, dh_loops=10, seed=None, out=sys.stdout):
    rng = random.Random(seed)
    rep = Report(out)
    G = c.AffineToR1(c.Gx, c.Gy)
    O = c.AffineToR1(c.Ox, c.Oy)

    # fields.py:366-399
    rep.check("inv-1271", GFp.mul(GFp.inv(13), 13) == 1)
    rep.check("invsqrt-1271", GFp.mul(13, GFp.sqr(GFp.invsqrt(13))) == 1)
    x23, x57 = (2, 3), (5, 7)
    for label, got, want in (("1+i", GFp2.add((1, 0), (0, 1)), (1, 1)), ("1*i", GFp2.mul((1, 0), (0, 1)), (0, 1)),
                             ("i*i", GFp2.mul((0, 1), (0, 1)), (p1271 - 1, 0)), ("add", GFp2.add(x23, x57), (7, 10)),
                             ("sub-pos", GFp2.sub(x57, x23), (3, 4)), ("sub-neg", GFp2.sub(x23, x57), (p1271 - 3, p1271 - 4)),
                             ("mul", GFp2.mul(x23, x57), (p1271 - 11, 29)), ("sqr", GFp2.sqr(x23), (p1271 - 5, 12)),
                             ("conj", GFp2.conj(x23), (2, GFp.neg(3))), ("inv-1271-2", GFp2.mul(x23, GFp2.inv(x23)), (1, 0))):
        rep.check(label, got == want, got)

    # curve4q.py:473-511
    rep.check("0-on-curve", c.PointOnCurve((c.Ox, c.Oy)))
    rep.check("G-on-curve", c.PointOnCurve((c.Gx, c.Gy)))
    rep.check("encode", bytes(c.encode(c.Gx, c.Gy)).hex() == KAT["Genc"])
    rep.check("decode", c.decode(bytearray(bytes.fromhex(KAT["Genc"]))) == (c.Gx, c.Gy))
    r1 = ((0, 1), (2, 0), (3, 4), (5, 0), (1, 6))
    td2 = GFp2.mul((2, 0), GFp2.mul(c.d, (5, 30)))
    rep.check("R1toR2", c.R1toR2(r1) == ((2, 1), (2, p1271 - 1), (6, 8), td2))
    rep.check("R1toR3", c.R1toR3(r1) == ((2, 1), (2, p1271 - 1), (3, 4), (5, 30)))
    rep.check("R2toR4", c.R2toR4(((2, 1), (2, p1271 - 1), (6, 8), td2)) == ((0, 2), (4, 0), (6, 8)))

    # curve4q.py:513-547
    A = (c.Gx, c.Gy, GFp2.one)
    for _ in range(loops):
        A = c.DBL(A)[:3]
    if loops == 1000:
        rep.point("double", A, KAT["doubleP"])
    rep.point("neutral-r", c.ADD(G, c.R1toR2(O)), G)
    rep.point("neutral-l", c.ADD(O, c.R1toR2(G)), G)
    Pt = G
    for _ in range(loops):
        Pt = c.ADD(Pt, c.R1toR2(Pt))
    rep.point("double-add", Pt, A)
    Pt, Q = c.DBL(G[:3]), c.R1toR2(G)
    for _ in range(loops):
        Pt = c.ADD(Pt, Q)
    if loops == 1000:
        rep.point("addition", Pt, KAT["P1000"])

    # curve4q.py:569-598, :675-704
    A2 = c.DBL(G)
    for name, mul, table in (("windowed", c.MUL_windowed, c.table_windowed), ("endo", c.MUL_endo, c.table_endo)):
        rep.point("mul-%s-*1" % name, mul(1, G), G)
        rep.point("mul-%s-*2" % name, mul(2, G), A2)
        Pt = G
        for m in kat_scalars(loops):
            Pt = mul(m, Pt)
        if loops == 1000:
            rep.point("mul-%s" % name, Pt, KAT["mulP"])
        T = table(G)
        rep.point("mul-%s-fixed-*1" % name, mul(1, G, table=T), G)
        rep.point("mul-%s-fixed-*2" % name, mul(2, G, table=T), A2)
        ms = [rng.getrandbits(256) for _ in range(10)]
        rep.check("mul-%s-fixed-rand" % name, all(mul(m, G, table=T) == mul(m, G) for m in ms))

    # curve4q.py:600-673
    Pt = Qt = G
    for _ in range(loops):
        Pt, Qt = c.phi(Pt), c.psi(Qt)
    if loops == 1000:
        rep.point("phi", Pt, KAT["phiP"])
        rep.point("psi", Qt, KAT["psiP"])
    for m, want in KAT["decompose"]:
        rep.check("decompose", c.decompose(m) == want)
    bad = 0
    for _ in range(min(loops, 200)):
        v = c.decompose(rng.getrandbits(256))
        signs, digits = c.recode(v)
        acc = [0, 0, 0, 0]
        for i in range(64, -1, -1):
            sg = 1 if signs[i] else -1
            acc = [2 * acc[0] + sg] + [2 * acc[j] + sg * ((digits[i] >> (j - 1)) & 1) for j in (1, 2, 3)]
        bad += acc != list(v)
    rep.check("recode", bad == 0, bad)

    # curve4q.py:706-778
    Gaff = (c.Gx, c.Gy)
    for name, dh in (("windowed", c.DH_windowed), ("endo", c.DH_endo)):
        Pt, bad = Gaff, 0
        for _ in range(dh_loops):
            m = rng.getrandbits(256)
            Q1 = dh(m, Pt)
            bad += Q1 != c.R1toAffine(c.MUL_windowed(392 * m, c.AffineToR1(Pt[0], Pt[1])))
            Pt = Q1
        rep.check("DH-%s-392" % name, bad == 0, bad)
        bad = 0
        for _ in range(dh_loops):
            a, b = rng.getrandbits(256), rng.getrandbits(256)
            bad += dh(a, dh(b, Gaff)) != dh(b, dh(a, Gaff))
        rep.check("DH-%s-symm" % name, bad == 0, bad)
    G392 = c.MUL_endo(392, G)
    for name, dh, T in (("windowed", c.DH_windowed, c.table_windowed(G392)), ("endo", c.DH_endo, c.table_endo(G392))):
        ms = [rng.getrandbits(256) for _ in range(dh_loops)]
        rep.check("DH-%s-fixed" % name, all(dh(m, Gaff, table=T) == dh(m, Gaff) for m in ms))
    for label, point in (("DH-reject-not-on-curve", ((0, 0), (0, 0))), ("DH-reject-392-torsion", KAT["P392"])):
        try:
            c.DH_endo(1, point)
            rep.check(label, False)
        except Exception:
            rep.check(label, True)
    return rep.failed


if __name__ == "__main__":
    sys.exit(1 if run() else 0)
