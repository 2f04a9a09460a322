"""GPU parity of the Diffie-Hellman wrapper (DH_core, curve4q.py:446-468) through the C ABI and
through the reference-shaped Python API, including the two rejection cases and their messages."""
import random

import numpy as np
import pytest

import curve4q_oracle as o
import oracle_c as oc
from conftest import unhex
from fourq_amd import codec

pytestmark = pytest.mark.gpu

G = (o.Gx, o.Gy)


def seeded_scalars(seed, n):
    rng = random.Random(seed)
    return np.frombuffer(rng.getrandbits(256 * n).to_bytes(32 * n, "little"), dtype="<u8").reshape(n, 4).copy()


def test_dh_golden(eng, golden):
    raw = golden("dh.json", raw=True)
    rows = unhex(raw["dh"])
    s = codec.pack_scalars([r[0] for r in rows])
    p = codec.pack_points([r[1] for r in rows], 2)
    out, st = eng.dh_endo(s, p)
    assert not st.any() and codec.unpack_points(out) == [r[2] for r in rows]
    out, st = eng.dh_windowed(s, p)
    assert not st.any() and codec.unpack_points(out) == [r[3] for r in rows]
    fx = unhex(raw["fixed"])
    s = codec.pack_scalars([r[0] for r in fx["rows"]])
    p = codec.pack_points([G] * len(fx["rows"]), 2)
    out, st = eng.dh_endo(s, p, codec.pack_table(fx["table_endo"]))
    assert not st.any() and codec.unpack_points(out) == [r[1] for r in fx["rows"]]
    out, st = eng.dh_windowed(s, p, codec.pack_table(fx["table_windowed"]))
    assert not st.any() and codec.unpack_points(out) == [r[2] for r in fx["rows"]]
    for a, b, ab in unhex(raw["exchange"]):
        out, st = eng.dh_exchange(codec.pack_scalars([a]), codec.pack_scalars([b]), codec.pack_point(G))
        assert st[0] == 0 and codec.unpack_fp2s(out[0]) == ab


def test_reference_api_and_errors(golden):
    """The drop-in module: same calls, same exceptions and messages as curve4q.py:446-468, :765-778."""
    from fourq_amd import curve4q as c
    raw = golden("dh.json", raw=True)
    for m, Pt, e, w in unhex(raw["dh"])[:4]:
        assert c.DH_endo(m, Pt) == e and c.DH_windowed(m, Pt) == w
        assert c.DH_core(m, Pt, c.MUL_endo) == e
    for m, Pt, msg in raw["reject"]:
        for dh in (c.DH_endo, c.DH_windowed):
            with pytest.raises(Exception) as ei:
                dh(int(m, 16), unhex(Pt))
            assert str(ei.value) == msg
    with pytest.raises(ValueError):
        c.MUL_endo(1, ((0, 0), (1, 0)))            # malformed R1 tuple -> unpacking error, as curve4q.py:407
    G1 = c.AffineToR1(c.Gx, c.Gy)
    assert c.R1toAffine(c.MUL_endo(1, G1)) == G and c.R1toAffine(c.mul(2, G1)) == o.R1toAffine(o.DBL(G1))
    T = c.table_endo(G1)
    assert T == o.table_endo(G1) and c.MUL_endo(12345, G1, table=T) == o.MUL_endo(12345, G1)
    assert c.MUL_windowed(-7, G1) == o.MUL_windowed(-7, G1)      # any integer: reduced mod N first (curve4q.py:217)
    assert c.PointOnCurve(G) and not c.PointOnCurve(((0, 0), (0, 0)))
    assert c.decompose(0x1234) == o.decompose(0x1234)
    assert tuple(map(list, c.recode(c.decompose(987654321)))) == tuple(map(list, o.recode(o.decompose(987654321))))
    assert c.GFp2.mul((2, 3), (5, 7)) == (o.P127 - 11, 29) and c.GFp.mul(c.GFp.inv(13), 13) == 1
    assert c.DBL(G1) == o.DBL(G1) and c.phi(G1) == o.phi(G1) and c.psi(G1) == o.psi(G1)
    a, b = 0xA5A5 << 200, 0x5A5A << 190
    assert c.dh_exchange(a, b) == o.dh_exchange(a, b)


def test_dh_batch_with_failures_vs_c_oracle(eng, golden):
    """4096 elements, some off-curve, some of small order: per-element status and zeroed outputs."""
    n = 4096
    s = seeded_scalars(40002, n)
    mid, st = eng.dh_endo(seeded_scalars(40003, n), np.repeat(codec.pack_point(G).reshape(1, 8), n, axis=0))
    assert not st.any()
    pts = mid.copy()
    p392 = codec.pack_point(unhex(golden("kat.json", raw=True)["P392"]))
    pts[5::97] = p392                              # 392-torsion -> neutral
    pts[11::101, 0] ^= np.uint64(1)                # perturbed x -> not on curve
    pts[17] = 0                                    # ((0,0),(0,0)) -> not on curve
    want, wst = oc.dh(oc.ENDO, s, pts)
    got, gst = eng.dh_endo(s, pts)
    assert np.array_equal(gst, wst) and np.array_equal(got, want)
    assert set(np.unique(gst)) == {0, 1, 2}
    want, wst = oc.dh(oc.WINDOWED, s[:1024], pts[:1024])
    got, gst = eng.dh_windowed(s[:1024], pts[:1024])
    assert np.array_equal(gst, wst) and np.array_equal(got, want)


def test_dh_symmetry_and_392_property(eng):
    """curve4q.py:709-741 at batch scale: DH(a, DH(b, G)) == DH(b, DH(a, G)); DH(m, P) == [392 m]P."""
    n = 8192
    a, b = seeded_scalars(61, n), seeded_scalars(62, n)
    g = codec.pack_point(G)
    ab, s1 = eng.dh_exchange(a, b, g)
    ba, s2 = eng.dh_exchange(b, a, g)
    assert not s1.any() and not s2.any() and np.array_equal(ab, ba)
    t392 = eng.table_endo(codec.pack_point(o.MUL_endo(392, o.AffineToR1(*G))))
    ab_fixed, s3 = eng.dh_exchange(a, b, g, table392=t392)        # fixed-base first half (curve4q.py:743-762)
    assert not s3.any() and np.array_equal(ab_fixed, ab)
    k = 512
    m392 = codec.pack_scalars([(392 * x) % o.N for x in codec.unpack_scalars(a[:k])])
    g1 = np.repeat(codec.pack_point(o.AffineToR1(*G)).reshape(1, 20), k, axis=0)
    direct = eng.prim("PT_R1TOAFFINE", eng.mul_windowed(m392, g1))
    dh, st = eng.dh_endo(a[:k], np.repeat(g.reshape(1, 8), k, axis=0))
    assert not st.any() and np.array_equal(dh, direct)


def test_full_size_cfg4_shard_exchange_symmetry(eng):
    """BASELINE.json config 4, one GPU's shard (2^19 of the 2^22 exchanges): DH(a, DH(b, G)) == DH(b, DH(a, G))
    for every pair, the fixed-base and variable-base first halves agree, and a 2^12 slice matches the C oracle."""
    n = 1 << 19
    a, b = seeded_scalars(40002, n), seeded_scalars(40003, n)
    g = codec.pack_point(G)
    t392 = eng.table_endo(codec.pack_point(o.MUL_endo(392, o.AffineToR1(*G))))
    ab, s1 = eng.dh_exchange(a, b, g, table392=t392)
    ba, s2 = eng.dh_exchange(b, a, g)
    assert not s1.any() and not s2.any() and np.array_equal(ab, ba)
    k = 1 << 12
    gk = np.repeat(g.reshape(1, 8), k, axis=0)
    mid, st = oc.dh(oc.ENDO, b[:k], gk)
    want, st2 = oc.dh(oc.ENDO, a[:k], mid)
    assert not st.any() and not st2.any() and np.array_equal(ab[:k], want)


def test_reference_self_test_sequence_on_the_gpu():
    """The reference's own self-tests (curve4q.py:473-790, fields.py:366-409) through the drop-in API: every line
    must read [PASS], including the 1000-step doubleP / P1000 / mulP / phiP / psiP chains."""
    import io
    from fourq_amd import selftest
    buf = io.StringIO()
    failed = selftest.run(loops=1000, dh_loops=4, seed=2026, out=buf)
    text = buf.getvalue()
    assert failed == 0 and "[FAIL]" not in text, text
    assert text.count("[PASS]") >= 45
    for label in ("double", "addition", "mul-windowed", "mul-endo", "phi", "psi", "encode", "decode", "DH-endo-symm", "DH-reject-392-torsion"):
        assert "[PASS] %s\n" % label in text


@pytest.mark.parametrize("group", [0, 2, 4, 8])
@pytest.mark.parametrize("n", [1, 7, 1000, 2049])
def test_batched_normalisation_groups(group, n, monkeypatch, golden):
    """R1toAffine with one inversion per `group` elements (Montgomery's trick): FOURQ_NORM_K forces the
    group size that large batches pick by themselves and FOURQ_SPLIT_MIN=512 sends n >= 512 down the prep + ladder route
    (which always defers; group 0 there means one inversion per element in normalize_kernel).  Rejected elements
    (off-curve, 392-torsion) sit inside the groups; every output and status must equal the C oracle's."""
    from fourq_amd import Engine
    monkeypatch.setenv("FOURQ_NORM_K", str(group))
    monkeypatch.setenv("FOURQ_SPLIT_MIN", "512")
    e = Engine(0)
    try:
        s = seeded_scalars(4242 + n, n)
        g = np.repeat(codec.pack_point(G).reshape(1, 8), n, axis=0)
        t392 = e.table_endo(codec.pack_point(o.MUL_endo(392, o.AffineToR1(*G))))
        pts, st = e.dh_endo(seeded_scalars(4343 + n, n), g, t392)              # fixed-base DH (LDS ladder)
        want, wst = oc.dh(oc.ENDO, seeded_scalars(4343 + n, n), g)
        assert not st.any() and np.array_equal(pts, want)
        pts = pts.copy()
        pts[3::7] = codec.pack_point(unhex(golden("kat.json", raw=True)["P392"]))   # -> neutral
        pts[5::11, 0] ^= np.uint64(1)                                                # -> not on curve
        for kind, fn in ((oc.ENDO, e.dh_endo), (oc.WINDOWED, e.dh_windowed)):       # variable base
            want, wst = oc.dh(kind, s, pts)
            got, gst = fn(s, pts)
            assert np.array_equal(gst, wst) and np.array_equal(got, want), (kind, group, n)
        comb = e.comb_table(codec.pack_point(o.MUL_endo(392, o.AffineToR1(*G))))
        s[::13] = 0                                                              # [0]B -> neutral inside a group
        want, wst = oc.dh(oc.ENDO, s, g)
        got, gst = e.comb_mul(s, comb)
        assert np.array_equal(gst, wst) and np.array_equal(got, want)
    finally:
        e.close()


def test_batched_normalisation_large_batches_pick_a_group(eng):
    """2^18 + 3 elements is >= 4 resident generations of fused lanes: group 4 by default.  Fixed-base DH, variable-base DH
    and the comb must all agree with the C oracle on a slice and with each other in full."""
    n = (1 << 18) + 3
    s = seeded_scalars(9191, n)
    g = np.repeat(codec.pack_point(G).reshape(1, 8), n, axis=0)
    g392 = codec.pack_point(o.MUL_endo(392, o.AffineToR1(*G)))
    fixed, st = eng.dh_endo(s, g, eng.table_endo(g392))
    assert not st.any()
    comb, st = eng.comb_mul(s, eng.comb_table(g392))
    assert not st.any() and np.array_equal(comb, fixed)
    var, st = eng.dh_endo(s, g)
    assert not st.any() and np.array_equal(var, fixed)
    for lo in (0, n - 2048):
        want, wst = oc.dh(oc.ENDO, s[lo:lo + 2048], g[:2048])
        assert np.array_equal(fixed[lo:lo + 2048], want)


def test_drop_in_key_generation_goes_through_the_comb():
    """`DH_endo(m, G)` / `DH_windowed(m, G)` on the curve's generator are key generation: the drop-in module computes them with the
    fixed-base comb of [392]G (draft-ladd-cfrg-4q.md:725-729 allows any method that agrees on all inputs).  Same affine points and
    the same exception as the reference's algorithm for edge scalars; FOURQ_COMB_KEYGEN=0 gives the general kernels."""
    from fourq_amd import curve4q as c
    N = o.N
    G = (o.Gx, o.Gy)
    before = c.combine_stats().get("keygen_comb", {"calls": 0})["calls"]
    ok = [1, 2, 3, N - 1, N + 1, 2 * N + 5, 1 << 255, (1 << 256) - 1, 0x1234567890abcdef1234567890abcdef1234567890abcdef1234567890abcdef]
    for m in ok:
        want = o.DH_endo(m, G)
        assert c.DH_endo(m, G) == want and c.DH_windowed(m, G) == want, hex(m)
    for m in (0, N, 2 * N):
        for dh in (c.DH_endo, c.DH_windowed):
            with pytest.raises(Exception) as ei:
                dh(m, G)
            assert str(ei.value) == "DH computation resulted in neutral point"
    assert c.DH_windowed(-5, G) == o.DH_windowed(-5, G)                     # any integer: reduced mod N first (curve4q.py:217)
    assert c.combine_stats()["keygen_comb"]["calls"] - before == 2 * len(ok) + 6 + 1
    saved = c._COMB_KEYGEN
    try:
        c._COMB_KEYGEN = False
        assert c.DH_endo(ok[-1], G) == o.DH_endo(ok[-1], G)                 # the general kernels agree
    finally:
        c._COMB_KEYGEN = saved
