#!/usr/bin/env python3
"""Regenerates tests/golden/*.json by running the REAL reference (build container only).

    python tests/golden/make_golden.py

Needs /root/reference (loaded in memory by oracle/ref_loader.py; nothing of it is copied).
Outputs are pure data: inputs and the reference's outputs, integers as hex strings.

Files written
  kat.json        literal known answers held by the reference's own self-tests
                  (curve4q.py:473-778, fields.py:366-403) together with their inputs
  field.json      seeded GFp / GFp2 vectors incl. edge values            
  group.json      DBL / ADD / ADD_core / R1toR2 / R1toR3 / R2toR4 / endomorphism pieces
  recode.json     decompose / recode / fixed-window digit vectors
  tables.json     table_windowed / table_endo for G and random points
  mul.json        MUL_windowed / MUL_endo raw R1 outputs (with/without table, edge scalars)
  dh.json         DH_windowed / DH_endo outputs and both rejection cases
  wire.json       encode / decode round trips, arbitrary strings, malformed inputs
  protocol.json   GFp.select / GFp2.select, and the protocol step encode(DH(m, decode(B))) composed from the reference's
                  own three functions (draft-ladd-cfrg-4q.md:707-723), incl. every way it fails
"""
import json
import os
import random
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "oracle"))
import ref_loader  # noqa: E402

F, C = ref_loader.load()
P = F.p1271


def hx(v):
    """ints -> hex strings, recursively through tuples/lists."""
    if isinstance(v, (bool, str)):
        return v
    if isinstance(v, dict):
        return {k: hx(e) for k, e in v.items()}
    if isinstance(v, int):
        return "%x" % v
    return [hx(e) for e in v]


def dump(name, obj):
    path = os.path.join(HERE, name)
    with open(path, "w") as fh:
        json.dump(obj, fh, separators=(",", ":"))
        fh.write("\n")
    print("%-14s %8d bytes" % (name, os.path.getsize(path)))


EDGE_FP = [0, 1, 2, P - 1, P - 2, (1 << 64) - 1, 1 << 64, (1 << 64) + 1, (1 << 126), (1 << 126) - 1, (1 << 96) - 1]


def rand_fp(rng):
    return rng.getrandbits(127) % P


def rand_f2(rng):
    return (rand_fp(rng), rand_fp(rng))


# ------------------------------------------------------------------------------ kat.json
def make_kat():
    kat = {
        "_source": "literal vectors in the reference self-tests; keys name the test",
        "p1271": P,
        "d": C.d,
        "N": C.N,
        "G": (C.Gx, C.Gy),
        "O": (C.Ox, C.Oy),
        # curve4q.py:517-522, :533-537  (1000 doublings of G; 1000 self-additions)
        "doubleP": ((0x2C3FD8822C82270FC9099C54855859D6, 0x4DA5B9E83AA7A1B2A7B3F6E2043E8E68),
                    (0x2001EB3A576883963EE089F0EB49AA14, 0x0FFDB0D761421F501FEE5617A7E954CD)),
        # curve4q.py:540-547  (2G + 1000*G)
        "P1000": ((0x3E243958590C4D906480B1EF0A151DB0, 0x5327AF7D84238CD0AA270F644A65D473),
                  (0x3EF69A49CB7E02375E06003D73C43EB1, 0x293EB1E26DD23B4E4E752648AC2EF0AB)),
        # curve4q.py:549-567  (1000 chained scalar mults; scalar rule in kat_scalars)
        "mul_chain_seed": [0x3AD457AB55456230, 0x3A8B3C2C6FD86E0C, 0x7E38F7C9CFBB9166, 0x0028FD6CBDA458F0],
        "mulP": ((0x257C122BBFC94A1BDFD2B477BD494BEF, 0x469BF80CB5B11F01769593547237C459),
                 (0x0901B3817C0E936C281C5067996F3344, 0x570B948EACACE2104FE8C429915F1245)),
        # curve4q.py:603-617
        "phiP": ((0x5550AAB9E7A620EED5B5A3061287DB16, 0x3E61EBB9A1CB0210EC321E6CF33610FC),
                 (0x5474BF8EC55603AE7E2851D5A8E83FB9, 0x5476093DBF8BF6BFA5077613491788D5)),
        "psiP": ((0x75AF54EDB41A2B93D8F3C8C24A2BC7E2, 0x065249F9EDE0C7984DE2466701F009A9),
                 (0x06DBB85BFFB7C21E1C6E119ADD608104, 0x060A30903424BF13FD234D6C4CFA3EC1)),
        # curve4q.py:623-634
        "decompose": [
            [0x92990788D66BF558052D112F5498111747B3E28C55984D43FED8C8822AD9F1A7,
             [0xA8EA3F673F711E51, 0xA08D1EAE0B9E071D, 0x55C8DF690050276F, 0x6396739DDA88830F]],
            [0x48E5CA2A675AB49CA214B884813935024B0C61EDC8D1305FE5230DF341623348,
             [0xA53EC4631945B875, 0x521C0BA1261C1934, 0x5C50CE912909185C, 0x93B3C70960B44BAD]],
            [0xAE20E251C36CFA5BE4D9F3D5A5EDFED305A1E8F7F6394D9BE58A15C4B0F1C5E9,
             [0xA621ADA9B3499C9F, 0x7CD17E0095E7AAE6, 0x6E8D23B5BD10BB43, 0x7F18C69F3025234C]],
            [0xB2C950ABC87A55442CC00F1E3AC38F81B7E95036FD191EA134FF616D9806E10C,
             [0x9B30A872EBEA83AF, 0x8F6C73350447C9C3, 0x72FDC76E3456D087, 0x6BA39BA159B0C13D]],
            [0x8E2958A1475ED70762340E9797788E0061F21FCEBD67889FDD4F4CE2B5F6B2DE,
             [0xBE8F3583A0934333, 0xAB45BF6D1BF80B37, 0x4A19FC5CFFE97809, 0x5EA3BAF1A1206442]],
        ],
        # curve4q.py:478
        "Genc": "87b2cb2b46a224b95a7820a19bee3f0e5c8b4c8444c3a74942020e63f84a1c6e",
        # curve4q.py:772-773  (a point of order dividing 392: DH must reject it)
        "P392": ((0x1318020702DE23BC3C9B73C751B4B192, 0x77AB39A7D8990C0A18E3C409FBD81A95),
                 (0x515854B6D19CC2DA1EA2B43B5121A22E, 0x763F89E129497361D74DFF5063E66682)),
        # curve4q.py:490-511 (test_reps): r1 -> r2, r3 ; r2 -> r4
        "reps": {
            "r1": ((0, 1), (2, 0), (3, 4), (5, 0), (1, 6)),
            "r2": ((2, 1), (2, P - 1), (6, 8), F.GFp2.mul((2, 0), F.GFp2.mul(C.d, (5, 30)))),
            "r3": ((2, 1), (2, P - 1), (3, 4), (5, 30)),
            "r4": ((0, 2), (4, 0), (6, 8)),
        },
        # fields.py:373-399 (test_GFp2) -- [op, inputs..., expected]
        "gfp2_literals": [
            ["add", (1, 0), (0, 1), (1, 1)],
            ["mul", (1, 0), (0, 1), (0, 1)],
            ["mul", (0, 1), (0, 1), (P - 1, 0)],
            ["add", (2, 3), (5, 7), (7, 10)],
            ["sub", (5, 7), (2, 3), (3, 4)],
            ["sub", (2, 3), (5, 7), (P - 3, P - 4)],
            ["mul", (2, 3), (5, 7), (P - 11, 29)],
            ["sqr", (2, 3), (P - 5, 12)],
            ["conj", (2, 3), (2, P - 3)],
        ],
    }
    # the reference asserts these itself; fail loudly if the loader ever drifts
    A = (C.Gx, C.Gy, F.GFp2.one)
    for _ in range(1000):
        A = C.DBL(A)[:3]
    zi = F.GFp2.inv(A[2])
    assert (F.GFp2.mul(A[0], zi), F.GFp2.mul(A[1], zi)) == kat["doubleP"]
    for m, v in kat["decompose"]:
        assert C.decompose(m) == v
    return kat


# ---------------------------------------------------------------------------- field.json
def make_field():
    rng = random.Random(1001)
    pairs = [(a, b) for a in EDGE_FP for b in EDGE_FP[:6]]
    pairs += [(rand_fp(rng), rand_fp(rng)) for _ in range(256 - len(pairs))]
    out = {"fp": [], "fp2": [], "fp_inv": [], "fp2_inv": []}
    for a, b in pairs:
        out["fp"].append(hx([a, b, F.GFp.add(a, b), F.GFp.sub(a, b), F.GFp.mul(a, b), F.GFp.sqr(a), F.GFp.neg(a)]))
    f2 = [((a, b), (b, a)) for a in EDGE_FP for b in EDGE_FP[:4]]
    f2 += [(rand_f2(rng), rand_f2(rng)) for _ in range(256 - len(f2))]
    for a, b in f2:
        out["fp2"].append(hx([a, b, F.GFp2.add(a, b), F.GFp2.sub(a, b), F.GFp2.mul(a, b), F.GFp2.sqr(a),
                              F.GFp2.neg(a), F.GFp2.conj(a)]))
    for a in EDGE_FP + [rand_fp(rng) for _ in range(53)]:
        out["fp_inv"].append(hx([a, F.GFp.inv(a), F.GFp.invsqrt(a)]))
    for a in [(x, y) for x in EDGE_FP[:4] for y in EDGE_FP[:4]] + [rand_f2(rng) for _ in range(48)]:
        out["fp2_inv"].append(hx([a, F.GFp2.inv(a)]))
    out["_layout"] = {"fp": "a,b,add,sub,mul,sqr(a),neg(a)", "fp2": "a,b,add,sub,mul,sqr(a),neg(a),conj(a)",
                      "fp_inv": "a,inv,invsqrt", "fp2_inv": "a,inv"}
    return out


def torsion_points(rng, count):
    """Projective (Z != 1) points of order N: raw R1 outputs of [k]G."""
    G = C.AffineToR1(C.Gx, C.Gy)
    return [C.MUL_endo(rng.getrandbits(256), G) for _ in range(count)]


# ---------------------------------------------------------------------------- group.json
def make_group():
    rng = random.Random(1002)
    pts = torsion_points(rng, 24)
    junk = [tuple(rand_f2(rng) for _ in range(5)) for _ in range(24)]  # formulas are total maps
    G = C.AffineToR1(C.Gx, C.Gy)
    O = C.AffineToR1(C.Ox, C.Oy)
    r1s = [G, O] + pts + junk
    out = {"r1": [], "add": [], "endo": [], "on_curve": []}
    for Pt in r1s:
        out["r1"].append(hx([Pt, C.DBL(Pt), C.R1toR2(Pt), C.R1toR3(Pt), C.R2toR4(C.R1toR2(Pt))]))
    for i, Pt in enumerate(r1s):
        Qt = r1s[(i * 7 + 3) % len(r1s)]
        q2 = C.R1toR2(Qt)
        out["add"].append(hx([Pt, q2, C.ADD(Pt, q2), C.ADD_core(C.R1toR3(Pt), q2)]))
    for Pt in [G] + pts[:15] + junk[:8]:
        t = C.tau(Pt[:3])
        out["endo"].append(hx([Pt, t, C.tau_dual(t), C.upsilon(t), C.chi(t), C.phi(Pt), C.psi(Pt)]))
    aff = [(C.Gx, C.Gy), (C.Ox, C.Oy), ((0, 0), (0, 0)), ((1, 0), (0, 0))]
    for Pt in pts[:8]:
        aff.append(C.R1toAffine(Pt))
    aff += [(rand_f2(rng), rand_f2(rng)) for _ in range(4)]
    for a in aff:
        out["on_curve"].append([hx(a), bool(C.PointOnCurve(a))])
    out["_layout"] = {"r1": "P,DBL,R1toR2,R1toR3,R2toR4(R1toR2)", "add": "P(R1),Q(R2),ADD,ADD_core(R1toR3(P),Q)",
                      "endo": "P,tau(P[:3]),tau_dual(t),upsilon(t),chi(t),phi(P),psi(P)", "on_curve": "affine,bool"}
    return out


EDGE_SCALARS = [0, 1, 2, 3, 15, 16, 17, C.N - 1, C.N, C.N + 1, 2 * C.N, 2 * C.N + 1, 1 << 255, (1 << 256) - 1,
                (1 << 256) - 2, (1 << 64) - 1, 1 << 64, (1 << 128) - 1, 1 << 192]


def windowed_digits(m):
    """Re-derives MUL_windowed's digit arrays (curve4q.py:216-226) with the reference's own arithmetic."""
    red = m % C.N
    if red % 2 == 0:
        red += C.N
    dg = []
    for _ in range(63):
        di = (red % 32) - 16
        dg.append(di)
        red = (red - di) // 16
    dg[62] = red
    return dg


# --------------------------------------------------------------------------- recode.json
def make_recode():
    rng = random.Random(1003)
    ms = EDGE_SCALARS + [rng.getrandbits(256) for _ in range(256 - len(EDGE_SCALARS))]
    out = {"decompose": [], "recode": [], "windowed": []}
    for m in ms:
        out["decompose"].append(hx([m, C.decompose(m)]))
    for m in ms[:96]:
        s, dg = C.recode(C.decompose(m))
        out["recode"].append([hx(m), "".join(map(str, s)), "".join(map(str, dg))])
    for m in ms[:96]:
        out["windowed"].append([hx(m), windowed_digits(m)])
    out["_layout"] = {"decompose": "m,[a1..a4]", "recode": "m,signs[0..64],digits[0..64]",
                      "windowed": "m,signed digits d[0..62] (curve4q.py:220-223)"}
    return out


# --------------------------------------------------------------------------- tables.json
def make_tables():
    rng = random.Random(1004)
    G = C.AffineToR1(C.Gx, C.Gy)
    out = []
    for Pt in [G] + torsion_points(rng, 5):
        out.append(hx([Pt, C.table_windowed(Pt), C.table_endo(Pt)]))
    return {"tables": out, "_layout": "P(R1),table_windowed(P)[8 R2],table_endo(P)[8 R2]"}


# ------------------------------------------------------------------------------ mul.json
def make_mul():
    rng = random.Random(1005)
    G = C.AffineToR1(C.Gx, C.Gy)
    negG = C.AffineToR1(F.GFp2.neg(C.Gx), C.Gy)
    pts = torsion_points(rng, 40)
    out = {"var": [], "fixed": [], "edge": []}
    for i, Pt in enumerate(pts):
        m = rng.getrandbits(256)
        out["var"].append(hx([m, Pt, C.MUL_endo(m, Pt), C.MUL_windowed(m, Pt)]))
    for base in (G, pts[0]):
        te, tw = C.table_endo(base), C.table_windowed(base)
        rows = []
        for _ in range(20):
            m = rng.getrandbits(256)
            rows.append(hx([m, C.MUL_endo(m, base, table=te), C.MUL_windowed(m, base, table=tw)]))
        out["fixed"].append({"P": hx(base), "table_endo": hx(te), "table_windowed": hx(tw), "rows": rows})
    for base in (G, negG, pts[1]):
        for m in EDGE_SCALARS:
            out["edge"].append(hx([m, base, C.MUL_endo(m, base), C.MUL_windowed(m, base)]))
    out["_layout"] = {"var": "m,P(R1),MUL_endo,MUL_windowed", "fixed.rows": "m,MUL_endo(table),MUL_windowed(table)",
                      "edge": "m,P(R1),MUL_endo,MUL_windowed"}
    return out


# ------------------------------------------------------------------------------- dh.json
def make_dh():
    rng = random.Random(1006)
    G = (C.Gx, C.Gy)
    out = {"dh": [], "fixed": [], "reject": []}
    Pt = G
    for _ in range(24):
        m = rng.getrandbits(256)
        e, w = C.DH_endo(m, Pt), C.DH_windowed(m, Pt)
        out["dh"].append(hx([m, Pt, e, w]))
        Pt = e
    for m in EDGE_SCALARS[1:8]:
        try:
            out["dh"].append(hx([m, G, C.DH_endo(m, G), C.DH_windowed(m, G)]))
        except Exception as exc:  # N, 2N... land on the neutral point
            out["reject"].append([hx(m), hx(G), str(exc)])
    G392 = C.MUL_endo(392, C.AffineToR1(C.Gx, C.Gy))  # curve4q.py:758
    te, tw = C.table_endo(G392), C.table_windowed(G392)
    rows = []
    for _ in range(12):
        m = rng.getrandbits(256)
        rows.append(hx([m, C.DH_endo(m, G, table=te), C.DH_windowed(m, G, table=tw)]))
    out["fixed"] = {"G392": hx(G392), "table_endo": hx(te), "table_windowed": hx(tw), "rows": rows}
    P392 = ((0x1318020702DE23BC3C9B73C751B4B192, 0x77AB39A7D8990C0A18E3C409FBD81A95),
            (0x515854B6D19CC2DA1EA2B43B5121A22E, 0x763F89E129497361D74DFF5063E66682))
    for m, bad in [(1, ((0, 0), (0, 0))), (1, P392), (12345, P392), (7, ((1, 2), (3, 4))), (C.N, G), (0, G)]:
        try:
            C.DH_endo(m, bad)
            raise SystemExit("reference accepted a point it should reject")
        except Exception as exc:
            msg = str(exc)
        try:
            C.DH_windowed(m, bad)
            raise SystemExit("reference accepted a point it should reject")
        except Exception as exc:
            assert str(exc) == msg
        out["reject"].append([hx(m), hx(bad), msg])
    # exchanges: DH(a, DH(b, G)) == DH(b, DH(a, G))           (curve4q.py:726-734)
    ex = []
    for _ in range(6):
        a, b = rng.getrandbits(256), rng.getrandbits(256)
        ab = C.DH_endo(a, C.DH_endo(b, G))
        assert ab == C.DH_endo(b, C.DH_endo(a, G))
        ex.append(hx([a, b, ab]))
    out["exchange"] = ex
    out["_layout"] = {"dh": "m,P(affine),DH_endo,DH_windowed", "fixed.rows": "m,DH_endo(G,table),DH_windowed(G,table)",
                      "reject": "m,P(affine),exception message", "exchange": "a,b,DH_endo(a,DH_endo(b,G))"}
    return out


# ----------------------------------------------------------------------------- wire.json
def make_wire():
    """encode / decode (curve4q.py:33-96): round trips, arbitrary strings, malformed inputs, and the
    AttributeError the reference raises on its t == 0 branch (curve4q.py:76-77)."""
    rng = random.Random(1007)
    G = C.AffineToR1(C.Gx, C.Gy)
    out = {"roundtrip": [], "strings": [], "malformed": []}
    pts = [(C.Gx, C.Gy), (F.GFp2.neg(C.Gx), C.Gy)]
    pts += [C.R1toAffine(C.MUL_endo(rng.getrandbits(256), G)) for _ in range(62)]
    for Pt in pts:
        enc = C.encode(Pt[0], Pt[1])
        assert C.decode(bytearray(enc)) == Pt
        out["roundtrip"].append([hx(Pt), bytes(enc).hex()])

    def outcome(raw):
        try:
            return ["ok", hx(C.decode(bytearray(raw)))]
        except Exception as exc:
            return [type(exc).__name__, str(exc)]

    strings = [bytes(C.encode(C.Ox, C.Oy)), bytes(31) + b"\x80", b"\x01" + bytes(15) + b"\x00" * 16,
               bytes([0xFF] * 15 + [0x7F] + [0] * 16), bytes([0] * 16 + [0xFF] * 15 + [0x7F]), bytes([0xFF] * 15 + [0x7F] + [0xFF] * 16)]
    for _ in range(250):
        raw = bytearray(rng.getrandbits(256).to_bytes(32, "little"))
        raw[15] &= 0x7F
        strings.append(bytes(raw))
    for raw in strings:
        out["strings"].append([raw.hex()] + outcome(raw))
    bad = [bytes(31), bytes(33), b"", bytes(15) + b"\x80" + bytes(16), bytes([0xFF] * 32)]
    for raw in bad:
        out["malformed"].append([raw.hex()] + outcome(raw))
    out["_layout"] = {"roundtrip": "affine point, encode() hex", "strings": "32-byte hex, 'ok' + decode() | exception type + message",
                      "malformed": "hex, exception type + message"}
    return out


# ----------------------------------------------------------------------------- protocol.json
def make_protocol():
    """GFp.select / GFp2.select (fields.py:59-64, :236-238) and the Diffie-Hellman step over the wire: decode the peer's
    32-byte key, DH_<kind> with the scalar, encode the result -- the reference's own functions composed
    (curve4q.py:49-96, :446-468, :41-46)."""
    rng = random.Random(1008)
    out = {"select": [], "select2": [], "dh_bytes": []}
    for _ in range(24):
        x, y = rng.getrandbits(127), rng.getrandbits(127)
        a, b = (rng.getrandbits(127), rng.getrandbits(127)), (rng.getrandbits(127), rng.getrandbits(127))
        for c in (0, 1):
            out["select"].append(hx([c, x, y, F.GFp.select(c, x, y)]))
            out["select2"].append(hx([c, a, b, F.GFp2.select(c, a, b)]))
    G1 = C.AffineToR1(C.Gx, C.Gy)
    P392 = ((0x1318020702DE23BC3C9B73C751B4B192, 0x77AB39A7D8990C0A18E3C409FBD81A95),
            (0x515854B6D19CC2DA1EA2B43B5121A22E, 0x763F89E129497361D74DFF5063E66682))
    keys = [bytes(C.encode(*C.R1toAffine(C.MUL_endo(rng.getrandbits(256), G1)))) for _ in range(20)]
    keys += [bytes(C.encode(C.Gx, C.Gy)), bytes(C.encode(*P392)), bytes(C.encode(C.Ox, C.Oy)), bytes(31) + b"\x80",
             bytes([0xFF] * 15 + [0x7F] + [0] * 16)]
    for _ in range(24):                                   # arbitrary strings: most decode to some point of the curve, some do not
        raw = bytearray(rng.getrandbits(256).to_bytes(32, "little"))
        raw[15] &= 0x7F
        keys.append(bytes(raw))
    scalars = [rng.getrandbits(256) for _ in keys]
    scalars[0], scalars[1], scalars[20] = 0, C.N, C.N     # neutral results from valid keys
    for m, B in zip(scalars, keys):
        row = [hx(m), B.hex()]
        for dh in (C.DH_endo, C.DH_windowed):
            try:
                row.append(["ok", bytes(C.encode(*dh(m, C.decode(bytearray(B))))).hex()])
            except Exception as exc:
                row.append([type(exc).__name__, str(exc)])
        out["dh_bytes"].append(row)
    out["_layout"] = {"select": "c,x,y,GFp.select(c,x,y)", "select2": "c,a,b,GFp2.select(c,a,b)",
                      "dh_bytes": "m, key hex, [ok, encode(DH_endo(m, decode(key))) hex | exception type, message], the same for DH_windowed"}
    return out


if __name__ == "__main__":
    if "--only-protocol" in sys.argv:                     # the other files are unchanged since round 1
        dump("protocol.json", make_protocol())
        raise SystemExit(0)
    dump("kat.json", hx(make_kat()))
    dump("field.json", make_field())
    dump("group.json", make_group())
    dump("recode.json", make_recode())
    dump("tables.json", make_tables())
    dump("mul.json", make_mul())
    dump("dh.json", make_dh())
    dump("wire.json", make_wire())
    dump("protocol.json", make_protocol())
