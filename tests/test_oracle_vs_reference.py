"""Cross-checks the CPU oracle against the REAL reference, loaded in memory by oracle/ref_loader.py.
Runs only where /root/reference is mounted (the build container); skipped on the GPU box."""
import random

import pytest

import curve4q_oracle as o
import ref_loader

pytestmark = pytest.mark.skipif(not ref_loader.available(), reason="reference not mounted")


@pytest.fixture(scope="module")
def ref():
    return ref_loader.load()


def test_reference_self_tests_pass(ref, capsys):
    """The loader must run the reference faithfully: its own self-tests print only [PASS]."""
    _, c = ref
    for fn in ("test_definitions", "test_reps", "test_core", "test_endo", "test_recoding", "test_mul_windowed", "test_mul_endo"):
        getattr(c, fn)()
    out = capsys.readouterr().out
    assert "[FAIL]" not in out and out.count("[PASS]") >= 20


def test_constants_identical(ref):
    f, c = ref
    assert (o.P127, o.d, o.N, o.Gx, o.Gy, o.Ox, o.Oy) == (f.p1271, c.d, c.N, c.Gx, c.Gy, c.Ox, c.Oy)
    assert o.ctau == c.ctau and o.ctaudual == c.ctaudual
    assert list(o.cphi) == [getattr(c, "cphi%d" % i) for i in range(10)]
    assert [o.cpsi[i] for i in (1, 2, 3, 4)] == [getattr(c, "cpsi%d" % i) for i in (1, 2, 3, 4)]
    assert [list(b) for b in o.BASIS] == [c.b1, c.b2, c.b3, c.b4] and list(o.ELL) == [c.L1, c.L2, c.L3, c.L4]
    assert list(o.OFFSET_C) == c.c and list(o.OFFSET_CP) == c.cp


def test_recoding_random(ref):
    _, c = ref
    rng = random.Random(11)
    edge = [0, 1, 2, c.N - 1, c.N, c.N + 1, (1 << 256) - 1, 1 << 255, (1 << 64) - 1, 1 << 64]
    for m in edge + [rng.getrandbits(256) for _ in range(20000)]:
        v = c.decompose(m)
        assert o.decompose(m) == v
        assert all(0 <= x < (1 << 64) for x in v)
    for m in edge + [rng.getrandbits(256) for _ in range(1500)]:
        rs, rd = c.recode(c.decompose(m))
        os_, od = o.recode(o.decompose(m))
        assert (list(rs), list(rd)) == (os_, od)
        assert od[64] <= 7


def test_field_random(ref):
    f, _ = ref
    rng = random.Random(12)
    for _ in range(3000):
        a, b = (rng.getrandbits(127) % o.P127, rng.getrandbits(127) % o.P127), (rng.getrandbits(127) % o.P127, rng.getrandbits(127) % o.P127)
        assert o.f2_mul(a, b) == f.GFp2.mul(a, b) and o.f2_sqr(a) == f.GFp2.sqr(a)
        assert o.f2_add(a, b) == f.GFp2.add(a, b) and o.f2_sub(a, b) == f.GFp2.sub(a, b)
    for _ in range(40):
        a = (rng.getrandbits(127) % o.P127, rng.getrandbits(127) % o.P127)
        assert o.f2_inv(a) == f.GFp2.inv(a) and o.fp_invsqrt(a[0]) == f.GFp.invsqrt(a[0])


def test_scalar_mul_and_dh_random(ref):
    _, c = ref
    rng = random.Random(13)
    P = c.AffineToR1(c.Gx, c.Gy)
    aff = (c.Gx, c.Gy)
    for i in range(60):
        m = rng.getrandbits(256)
        e = c.MUL_endo(m, P)
        assert o.MUL_endo(m, P) == e and o.MUL_windowed(m, P) == c.MUL_windowed(m, P)
        if i % 6 == 0:
            assert o.table_endo(P) == c.table_endo(P) and o.table_windowed(P) == c.table_windowed(P)
            assert o.DH_endo(m, aff) == c.DH_endo(m, aff) and o.DH_windowed(m, aff) == c.DH_windowed(m, aff)
            aff = o.DH_endo(m, aff)
        P = e


def test_wire_random(ref):
    _, c = ref
    rng = random.Random(14)
    for _ in range(150):
        raw = bytearray(rng.getrandbits(256).to_bytes(32, "little"))
        raw[15] &= 0x7F
        try:
            want = ("ok", c.decode(bytearray(raw)))
        except Exception as exc:
            want = (type(exc).__name__, str(exc))
        try:
            got = ("ok", o.decode(bytes(raw)))
        except Exception as exc:
            got = (type(exc).__name__, str(exc))
        assert got == want


def test_mirror_module_has_every_public_name_of_the_reference(ref):
    """A user who swaps `import curve4q` for `from fourq_amd import curve4q` must find every module-level name
    (functions and constants; the reference's own test functions and its `getrandbits` import aside), and every
    constant must hold the same value.  Importing the mirror needs no GPU."""
    f, c = ref
    from fourq_amd import curve4q as m
    from fourq_amd import fields as mf
    skip = {"getrandbits", "test"} | {n for n in dir(c) if n.startswith("test_")}
    public = [n for n in dir(c) if not n.startswith("_") and n not in skip]
    assert [n for n in public if not hasattr(m, n)] == []
    for n in public:
        v = getattr(c, n)
        if isinstance(v, (int, tuple, list)):
            assert getattr(m, n) == v, n
    for cls in ("GFp", "GFp2"):
        want = [n for n in dir(getattr(f, cls)) if not n.startswith("_") and n not in ("A", "S", "M", "I", "ctr", "ctr_reset", "ctr_enabled")]   # op counters: not reproduced
        have = dir(getattr(mf, cls))
        assert [n for n in want if n not in have] == [], cls


def test_gfp2_invsqrt_as_written(ref):
    """GFp2.invsqrt (fields.py:202-230) is dead code in the reference; the restatement follows it to the letter."""
    f, _ = ref
    rng = random.Random(15)
    cases = [(rng.getrandbits(127) % f.p1271, rng.getrandbits(127) % f.p1271) for _ in range(40)]
    cases += [(rng.getrandbits(127) % f.p1271, 0) for _ in range(20)] + [(4, 0), (1, 0), (2, 0)]
    for a in cases:
        assert o.GFp2.invsqrt(a) == f.GFp2.invsqrt(a), a
