"""Pins the C oracle (oracle/fourq_oracle.c) against the golden vectors and the Python oracle."""
import random

import numpy as np

import curve4q_oracle as o
import oracle_c as oc
from fourq_amd import codec

G1 = o.AffineToR1(o.Gx, o.Gy)


def test_tables(golden):
    for Pt, tw, te in golden("tables.json")["tables"]:
        assert codec.unpack_table(oc.table(oc.WINDOWED, codec.pack_point(Pt))) == list(tw)
        assert codec.unpack_table(oc.table(oc.ENDO, codec.pack_point(Pt))) == list(te)


def test_mul_vectors(golden):
    g = golden("mul.json")
    rows = g["var"] + g["edge"]
    s = codec.pack_scalars([r[0] for r in rows])
    p = codec.pack_points([r[1] for r in rows], 5)
    assert codec.unpack_points(oc.mul(oc.ENDO, s, p)) == [r[2] for r in rows]
    assert codec.unpack_points(oc.mul(oc.WINDOWED, s, p)) == [r[3] for r in rows]
    for blk in g["fixed"]:
        s = codec.pack_scalars([r[0] for r in blk["rows"]])
        assert codec.unpack_points(oc.mul(oc.ENDO, s, None, codec.pack_table(blk["table_endo"]))) == [r[1] for r in blk["rows"]]
        assert codec.unpack_points(oc.mul(oc.WINDOWED, s, None, codec.pack_table(blk["table_windowed"]))) == [r[2] for r in blk["rows"]]


def test_dh_vectors(golden):
    from conftest import unhex
    raw_all = golden("dh.json", raw=True)
    g = {"dh": unhex(raw_all["dh"]), "fixed": unhex(raw_all["fixed"])}
    s = codec.pack_scalars([r[0] for r in g["dh"]])
    p = codec.pack_points([r[1] for r in g["dh"]], 2)
    out, st = oc.dh(oc.ENDO, s, p)
    assert not st.any() and codec.unpack_points(out) == [r[2] for r in g["dh"]]
    out, st = oc.dh(oc.WINDOWED, s, p)
    assert not st.any() and codec.unpack_points(out) == [r[3] for r in g["dh"]]
    fx = g["fixed"]
    s = codec.pack_scalars([r[0] for r in fx["rows"]])
    p = codec.pack_points([(o.Gx, o.Gy)] * len(fx["rows"]), 2)
    out, st = oc.dh(oc.ENDO, s, p, codec.pack_table(fx["table_endo"]))
    assert not st.any() and codec.unpack_points(out) == [r[1] for r in fx["rows"]]
    raw = raw_all["reject"]
    for m, Pt, msg in raw:
        out, st = oc.dh(oc.ENDO, codec.pack_scalars([int(m, 16)]), codec.pack_points([unhex(Pt)], 2))
        assert st[0] == (1 if msg == "Point not on curve" else 2) and not out.any()


def test_random_vs_python_oracle():
    rng = random.Random(4242)
    ms = [rng.getrandbits(256) for _ in range(24)]
    pts = [G1]
    for m in ms[:-1]:
        pts.append(o.MUL_endo(m, pts[-1]))
    s, p = codec.pack_scalars(ms), codec.pack_points(pts, 5)
    assert codec.unpack_points(oc.mul(oc.ENDO, s, p)) == [o.MUL_endo(m, P) for m, P in zip(ms, pts)]
    assert codec.unpack_points(oc.mul(oc.WINDOWED, s, p)) == [o.MUL_windowed(m, P) for m, P in zip(ms, pts)]
    dec = oc.decompose(s)
    assert [list(map(int, r)) for r in dec] == [o.decompose(m) for m in ms]


def test_chain_kat(golden):
    from conftest import unhex
    kat = golden("kat.json", raw=True)
    A = codec.pack_point(G1).reshape(1, 20)
    for m in o.kat_scalars(1000):
        A = oc.mul(oc.ENDO, codec.pack_scalars([m]), A)
    P = codec.unpack_fp2s(A[0])
    assert o.R1toAffine(P) == unhex(kat["mulP"])
