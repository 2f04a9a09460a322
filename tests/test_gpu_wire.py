"""GPU parity of the 32-byte point encoding (encode / decode, curve4q.py:33-96)."""
import random

import numpy as np
import pytest

import curve4q_oracle as o
from conftest import unhex
from fourq_amd import _lib, codec

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from fourq_amd import Engine
    e = Engine(0)
    yield e
    e.close()


def test_wire_golden(eng, golden):
    w = golden("wire.json", raw=True)
    pts = [unhex(r[0]) for r in w["roundtrip"]]
    enc = eng.encode(codec.pack_points(pts, 2))
    assert [bytes(r).hex() for r in enc] == [r[1] for r in w["roundtrip"]]
    out, st = eng.decode(enc)
    assert not st.any() and codec.unpack_points(out) == pts
    rows = w["strings"]
    raw = np.frombuffer(b"".join(bytes.fromhex(r[0]) for r in rows), dtype=np.uint8).reshape(-1, 32)
    out, st = eng.decode(raw)
    want = {"Malformed point: reserved bit is not zero": _lib.DECODE_RESERVED_BIT, "Point not on curve": _lib.DECODE_NOT_ON_CURVE,
            "type object 'GFp' has no attribute 'two'": _lib.DECODE_REF_ATTRIBUTE_ERROR}
    for r, got, s in zip(rows, out, st):
        if r[1] == "ok":
            assert s == 0 and codec.unpack_fp2s(got) == unhex(r[2])
        else:
            assert s == want[r[2]] and not got.any()
    assert set(st) == {0, 1, 2, 3}


def test_reference_shaped_wire_api(golden):
    from fourq_amd import curve4q as c
    w = golden("wire.json", raw=True)
    kat = golden("kat.json", raw=True)
    assert bytes(c.encode(c.Gx, c.Gy)).hex() == kat["Genc"]                      # curve4q.py:478-481
    assert c.decode(bytearray(bytes.fromhex(kat["Genc"]))) == (c.Gx, c.Gy)        # curve4q.py:484-485
    for row in w["malformed"] + [r for r in w["strings"] if r[1] != "ok"][:12]:
        with pytest.raises(Exception) as ei:
            c.decode(bytearray(bytes.fromhex(row[0])))
        assert (type(ei.value).__name__, str(ei.value)) == (row[1], row[2])
    assert c.GFp.mul(13, c.GFp.sqr(c.GFp.invsqrt(13))) == 1                       # fields.py:369-370
    assert c.sign(c.Gx) == o.sign(o.Gx)


def test_decode_random_strings_vs_oracle(eng):
    rng = random.Random(606)
    n = 4096
    raw = np.frombuffer(rng.getrandbits(256 * n).to_bytes(32 * n, "little"), dtype=np.uint8).reshape(n, 32).copy()
    raw[::7, 15] |= 0x80                                   # some reserved-bit violations
    out, st = eng.decode(raw)
    for i in range(0, n, 5):
        try:
            want, ws = o.decode(bytes(raw[i])), 0
        except AttributeError:
            want, ws = None, 3
        except Exception as exc:
            want, ws = None, 1 if "reserved" in str(exc) else 2
        assert st[i] == ws
        if ws == 0:
            assert codec.unpack_fp2s(out[i]) == want
    good = out[st == 0]
    assert np.array_equal(eng.encode(good), raw[st == 0])  # decode -> encode is the identity on valid encodings


def test_dh_over_the_wire(eng):
    """Both parties of draft-ladd-cfrg-4q's exchange, keys and secrets as 32-byte strings."""
    n = 2048
    rng = random.Random(707)
    a = np.frombuffer(rng.getrandbits(256 * n).to_bytes(32 * n, "little"), dtype="<u8").reshape(n, 4).copy()
    b = np.frombuffer(rng.getrandbits(256 * n).to_bytes(32 * n, "little"), dtype="<u8").reshape(n, 4).copy()
    genc = np.repeat(eng.encode(codec.pack_point((o.Gx, o.Gy)).reshape(1, 8)), n, axis=0)
    pa, s1 = eng.dh_bytes(a, genc)
    pb, s2 = eng.dh_bytes(b, genc)
    kab, s3 = eng.dh_bytes(a, pb)
    kba, s4 = eng.dh_bytes(b, pa)
    assert not (s1.any() or s2.any() or s3.any() or s4.any()) and np.array_equal(kab, kba)
    m, bm = codec.unpack_scalars(a[:1])[0], codec.unpack_scalars(b[:1])[0]
    want = o.DH_endo(m, o.DH_endo(bm, (o.Gx, o.Gy)))
    assert bytes(kab[0]) == bytes(o.encode(*want))
    bad = genc.copy(); bad[0, 0] ^= 1
    out, st = eng.dh_bytes(a[:4], bad[:4])
    assert st[0] in (16 + 2, 0) and not st[1:].any()
