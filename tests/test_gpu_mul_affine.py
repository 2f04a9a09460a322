"""MUL_* with affine and with encoded I/O (SURVEY 8(d) "affine-only I/O variant"; VERDICT r3 missing 2 / item 3):
fourq_mul_{endo,windowed}_affine_batch[_dev] = R1toAffine(MUL_*(m, AffineToR1(P))), fourq_mul_*_bytes_batch[_dev] = encode(.) of it
on decode(B) -- against the Python oracle on the golden rows and the edge scalars, against the C oracle on whole batches, both
selection modes, host-array and device-resident flavours, every route size class."""
import random

import numpy as np
import pytest

import curve4q_oracle as o
import oracle_c as oc
from bench import seeded_scalars
from fourq_amd import _lib, codec

pytestmark = pytest.mark.gpu

G = (o.Gx, o.Gy)
G1 = o.AffineToR1(o.Gx, o.Gy)
EDGE = [0, 1, 2, o.N - 1, o.N, o.N + 1, 2 * o.N, 1 << 255, (1 << 256) - 1]


def _affine_points(n, seed):
    """n affine N-torsion points [k_i]G (C oracle: fixed-base DH_endo does the cofactor too; plain MUL + R1toAffine here)."""
    ks = seeded_scalars(seed, n)
    r1 = oc.mul(oc.ENDO, ks, None, oc.table(oc.ENDO, codec.pack_point(G1)))
    return np.array([codec.pack_point(o.R1toAffine(P)) for P in codec.unpack_points(r1)], dtype=np.uint64)


def _want_affine(kind, scalars, aff):
    """Through the C oracle: MUL on the lifted points, then R1toAffine by the Python oracle (exact, small n) or by DH-free division."""
    n = len(scalars)
    r1 = np.zeros((n, 20), dtype=np.uint64)
    r1[:, 0:8] = aff
    r1[:, 8] = 1
    r1[:, 12:20] = aff
    out = oc.mul(oc.ENDO if kind == "endo" else oc.WINDOWED, scalars, r1)
    return out


def test_affine_flavour_on_edge_scalars_and_golden_points(eng, golden):
    rng = random.Random(4401)
    pts = [G, (o.Gx, o.GFp2.neg(o.Gy))] + [o.R1toAffine(r[1]) for r in golden("mul.json")["var"][:6]]
    ms = EDGE + [rng.getrandbits(256) for _ in range(7)]
    rows = [(m, P) for m in ms for P in pts]
    s = codec.pack_scalars([r[0] for r in rows])
    p = codec.pack_points([r[1] for r in rows], 2)
    for kind, mul in (("endo", o.MUL_endo), ("windowed", o.MUL_windowed)):
        got = eng.mul_affine(s, p, kind=kind)
        want = [o.R1toAffine(mul(m, o.AffineToR1(*P))) for m, P in rows]
        assert codec.unpack_points(got) == want, kind
        enc_in = np.frombuffer(b"".join(bytes(o.encode(*P)) for _, P in rows), dtype=np.uint8).reshape(-1, 32)
        out, st = eng.mul_bytes(s, enc_in, kind=kind)
        assert not st.any() and [bytes(r) for r in out] == [bytes(o.encode(*Q)) for Q in want], kind


def test_bytes_flavour_reports_undecodable_points(eng, golden):
    w = golden("wire.json", raw=True)
    rows = w["strings"][:64]
    raw = np.frombuffer(b"".join(bytes.fromhex(r[0]) for r in rows), dtype=np.uint8).reshape(-1, 32)
    s = seeded_scalars(4402, len(rows))
    out, st = eng.mul_bytes(s, raw)
    code = {"Malformed point: reserved bit is not zero": _lib.DECODE_RESERVED_BIT, "Point not on curve": _lib.DECODE_NOT_ON_CURVE,
            "type object 'GFp' has no attribute 'two'": _lib.DECODE_REF_ATTRIBUTE_ERROR}
    ms = codec.unpack_scalars(s)
    for r, m, got, v in zip(rows, ms, out, st):
        if r[1] == "ok":
            P = o.decode(bytearray(bytes.fromhex(r[0])))
            assert v == 0 and bytes(got) == bytes(o.encode(*o.R1toAffine(o.MUL_endo(m, o.AffineToR1(*P)))))
        else:
            assert v == _lib.BYTES_DECODE_BASE + code[r[2]] and not got.any()


@pytest.mark.parametrize("kind", ["endo", "windowed"])
def test_affine_flavour_whole_batches_vs_c_oracle(eng, kind):
    """Sizes on every route (four / two lanes per element, one fused generation, a generation and a remainder); host arrays and
    device-resident arrays give the same words; the raw-R1 entry point on the lifted points normalises to the same result."""
    import torch
    lanes = eng.lanes
    big = lanes + 300
    aff = _affine_points(big, 4403)
    s = seeded_scalars(4404, big)
    want_r1 = _want_affine(kind, s, aff)
    # R1toAffine of the C oracle's R1 rows: through the GPU's own DH-free normaliser is what is under test, so use the Python oracle on a sample
    # and the projective identity X_want * Z == x * Z ... for the rest: x = X/Z  <=>  the encode() of both agree; here: compare with
    # affine words computed by the device from the ORACLE's R1 rows via the raw-R1 path's inverse -- instead, check exactly on a sample
    for m in (1, 63, 4097, lanes // 4 + 1, lanes // 2 + 1, lanes, big):
        got = eng.mul_affine(s[:m], aff[:m], kind=kind)
        idx = sorted(set([0, m - 1] + random.Random(m).sample(range(m), min(m, 24))))
        want = [o.R1toAffine(P) for P in codec.unpack_points(want_r1[idx])]
        assert codec.unpack_points(got[idx]) == want, (kind, m)
        dev = torch.device("cuda", 0)
        sd, pd = (torch.from_numpy(a.view(np.int64)).to(dev) for a in (np.ascontiguousarray(s[:m]), np.ascontiguousarray(aff[:m])))
        od = torch.empty((m, 8), dtype=torch.int64, device=dev)
        eng.mul_affine_dev(sd, pd, od, m, kind=kind)
        eng.sync()
        assert np.array_equal(od.cpu().numpy().view(np.uint64), got), (kind, m)
    # every element of the largest batch: Z_oracle * x_gpu == X_oracle etc. would need field arithmetic here; the encode route does it on
    # the device against the oracle's DH-style division instead: encode(affine) must equal encode of the Python oracle on a second sample
    got = eng.mul_affine(s, aff, kind=kind)
    enc = eng.encode(got)
    for i in random.Random(7).sample(range(big), 16):
        assert bytes(enc[i]) == bytes(o.encode(*o.R1toAffine(codec.unpack_points(want_r1[i:i + 1])[0])))


def test_bytes_flavour_whole_batch_consistency(eng):
    """mul_bytes == encode(mul_affine(decode())) element for element at a size past one generation, device and host flavours."""
    import torch
    n = eng.lanes + 77
    aff = _affine_points(n, 4405)
    s = seeded_scalars(4406, n)
    enc_in = eng.encode(aff)
    out, st = eng.mul_bytes(s, enc_in)
    assert not st.any()
    assert np.array_equal(out, eng.encode(eng.mul_affine(s, aff)))
    dev = torch.device("cuda", 0)
    sd = torch.from_numpy(np.ascontiguousarray(s).view(np.int64)).to(dev)
    bd = torch.from_numpy(np.ascontiguousarray(enc_in)).to(dev)
    od = torch.empty((n, 32), dtype=torch.uint8, device=dev)
    std = torch.empty(n, dtype=torch.uint8, device=dev)
    eng.mul_bytes_dev(sd, bd, od, std, n)
    eng.sync()
    assert np.array_equal(od.cpu().numpy(), out) and not std.cpu().numpy().any()


def test_batched_normaliser_keeps_elements_independent(eng):
    """Round 5: past two generations R1toAffine behind a MUL_* shares ONE inversion among four elements of a lane (lower_kernel<4>).
    MUL_* accepts any pair of field elements (the reference never checks), and the all-zero point -- what a failed decode is lifted
    to, or what a caller may simply pass -- comes out with Z = 0: such an element must get (0, 0), as the reference's conj(0) * 0^(p-2)
    gives, and must not touch the three elements that share its inversion.  EVERY element against the C oracle, affine and encoded I/O,
    garbage at the head of a group, inside one, in all four slots of one lane, and at the ragged end.  The device-resident calls hand the
    whole batch to one launch (four elements per lane); the host-array calls cut it into chunks of one generation (one element per lane)."""
    import torch
    dev = torch.device("cuda", 0)

    def to_dev(a):
        a = np.ascontiguousarray(a)
        return torch.from_numpy(a.view(np.int64) if a.dtype == np.uint64 else a).to(dev)
    lanes = eng.lanes
    n = 3 * lanes + 41
    T = (n + 3) // 4                                            # lane t owns t, t + T, t + 2T, t + 3T
    rng = random.Random(4410)
    aff = oc.r1_to_affine(oc.mul(oc.ENDO, seeded_scalars(4407, n), None, oc.table(oc.ENDO, codec.pack_point(G1))))
    s = seeded_scalars(4408, n)
    zero_at = [0, T, 5, 5 + T, 5 + 2 * T, 5 + 3 * T, 77 + 2 * T, n - 1, 3 * T - 1] + rng.sample(range(n), 40)
    junk_at = [9, 9 + T] + rng.sample(range(n), 40)
    aff = aff.copy()
    for i in junk_at:                                           # arbitrary field elements: not a point of the curve
        aff[i] = np.frombuffer(rng.getrandbits(512).to_bytes(64, "little"), dtype="<u8") & np.uint64(0x7FFFFFFFFFFFFFFF)
    for i in zero_at:
        aff[i] = 0
    lifted = np.zeros((n, 20), dtype=np.uint64)
    lifted[:, 0:8] = aff
    lifted[:, 8] = 1
    lifted[:, 12:20] = aff
    # the oracle reduces words >= p as the device does (fe_unpack / % p1271): keep the junk below p so that both read the same value
    for kind, okind in (("endo", oc.ENDO), ("windowed", oc.WINDOWED)):
        want = oc.r1_to_affine(oc.mul(okind, s, lifted))
        assert not want[zero_at].any()                          # (0, 0), by the oracle too
        od = torch.empty((n, 8), dtype=torch.int64, device=dev)
        eng.mul_affine_dev(to_dev(s), to_dev(aff), od, n, kind=kind)
        eng.sync()
        for got in (od.cpu().numpy().view(np.uint64), eng.mul_affine(s, aff, kind=kind)):
            bad = np.flatnonzero((got != want).any(axis=1))
            assert bad.size == 0, (kind, bad[:8], [int(b) % T for b in bad[:8]])
    # encoded I/O: undecodable strings in the same positions (a reserved bit), everything else decodable
    ok_aff = oc.r1_to_affine(oc.mul(oc.ENDO, seeded_scalars(4407, n), None, oc.table(oc.ENDO, codec.pack_point(G1))))
    keys = oc.encode(ok_aff).copy()
    for i in zero_at:
        keys[i, 15] |= 0x80
    lifted_ok = np.zeros((n, 20), dtype=np.uint64)
    lifted_ok[:, 0:8] = ok_aff
    lifted_ok[:, 8] = 1
    lifted_ok[:, 12:20] = ok_aff
    want_enc = oc.encode(oc.r1_to_affine(oc.mul(oc.ENDO, s, lifted_ok))).copy()
    want_st = np.zeros(n, dtype=np.uint8)
    want_enc[zero_at] = 0
    want_st[zero_at] = 16 + 1
    got, st = eng.mul_bytes(s, keys)
    assert np.array_equal(st, want_st) and np.array_equal(got, want_enc)
    oe, ost = torch.empty((n, 32), dtype=torch.uint8, device=dev), torch.empty(n, dtype=torch.uint8, device=dev)
    eng.mul_bytes_dev(to_dev(s), to_dev(keys), oe, ost, n)
    eng.sync()
    assert np.array_equal(ost.cpu().numpy(), want_st) and np.array_equal(oe.cpu().numpy(), want_enc)
