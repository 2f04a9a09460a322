#!/usr/bin/env python3
"""One-off parity soak at the largest BASELINE.json size (2^22 elements): every output of every batched entry
point against the C oracle.  Test infrastructure (uses oracle/); run on the GPU box, result kept in profiles/."""
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import curve4q_oracle as o
import oracle_c as oc
from fourq_amd import Engine, codec

LG = int(sys.argv[1]) if len(sys.argv) > 1 else 22
n = 1 << LG


def scalars(seed):
    return np.frombuffer(random.Random(seed).getrandbits(256 * n).to_bytes(32 * n, "little"), dtype="<u8").reshape(n, 4).copy()


G1 = o.AffineToR1(o.Gx, o.Gy)
g1 = codec.pack_point(G1)
with Engine(0) as eng:
    te, tw = eng.table_endo(g1), eng.table_windowed(g1)
    s = scalars(1)
    pts = eng.mul_endo_fixed(scalars(2), te)
    checks = [
        ("MUL_endo variable base", lambda: eng.mul_endo(s, pts), lambda: oc.mul(oc.ENDO, s, pts)),
        ("MUL_windowed variable base", lambda: eng.mul_windowed(s, pts), lambda: oc.mul(oc.WINDOWED, s, pts)),
        ("MUL_endo fixed base", lambda: eng.mul_endo_fixed(s, te), lambda: oc.mul(oc.ENDO, s, None, te)),
        ("MUL_windowed fixed base", lambda: eng.mul_windowed_fixed(s, tw), lambda: oc.mul(oc.WINDOWED, s, None, tw)),
    ]
    for name, gpu, cpu in checks:
        t0 = time.time(); a = gpu(); t1 = time.time(); b = cpu(); t2 = time.time()
        print("%-28s n=2^%d  GPU %.2fs (PCIe incl.)  C oracle %.1fs  identical=%s" % (name, LG, t1 - t0, t2 - t1, np.array_equal(a, b)), flush=True)
        assert np.array_equal(a, b)
    flags = (np.frombuffer(random.Random(5).getrandbits(8 * n).to_bytes(n, "little"), dtype=np.uint8) & 1).copy()
    t0 = time.time(); a = eng.mul_endo_mixed(s, pts, flags, te); t1 = time.time()
    b = np.where(flags.reshape(-1, 1) != 0, oc.mul(oc.ENDO, s, pts), oc.mul(oc.ENDO, s, None, te))
    print("%-28s n=2^%d  GPU %.2fs (PCIe incl.)  identical=%s" % ("MUL_endo mixed 50/50", LG, t1 - t0, np.array_equal(a, b)), flush=True)
    assert np.array_equal(a, b)
    aff, st = eng.dh_endo(scalars(3), np.repeat(codec.pack_point((o.Gx, o.Gy)).reshape(1, 8), n, axis=0), te if False else None)
    assert not st.any()
    for name, gpu, cpu in (("DH_endo variable base", lambda: eng.dh_endo(s, aff), lambda: oc.dh(oc.ENDO, s, aff)),
                           ("DH_windowed variable base", lambda: eng.dh_windowed(s, aff), lambda: oc.dh(oc.WINDOWED, s, aff))):
        t0 = time.time(); a, sa = gpu(); t1 = time.time(); b, sb = cpu(); t2 = time.time()
        ok = np.array_equal(a, b) and np.array_equal(sa, sb)
        print("%-28s n=2^%d  GPU %.2fs (PCIe incl.)  C oracle %.1fs  identical=%s" % (name, LG, t1 - t0, t2 - t1, ok), flush=True)
        assert ok
    g392 = codec.pack_point(o.MUL_endo(392, G1))
    comb = eng.comb_table(g392)
    a, sa = eng.comb_mul(s, comb)
    gaff = np.repeat(codec.pack_point((o.Gx, o.Gy)).reshape(1, 8), n, axis=0)
    b, sb = oc.dh(oc.ENDO, s, gaff)
    ok = np.array_equal(a, b) and np.array_equal(sa, sb)
    print("%-28s n=2^%d  identical=%s" % ("comb keygen == DH_endo(m, G)", LG, ok), flush=True)
    assert ok
    enc = eng.encode(a)
    dec, dst = eng.decode(enc)
    ok = not dst.any() and np.array_equal(dec, a)
    print("%-28s n=2^%d  identical=%s" % ("decode(encode(P)) == P", LG, ok), flush=True)
    assert ok
    # protocol-level calls (r02): keys and secrets as 32-byte strings, both halves of an exchange chained on the device
    k2 = scalars(4)
    t0 = time.time(); sec, sst = eng.dh_bytes(k2, enc); t1 = time.time()
    want, wst = oc.dh(oc.ENDO, k2, b)
    ok = not sst.any() and not wst.any() and np.array_equal(sec, eng.encode(want))
    print("%-28s n=2^%d  GPU %.2fs (PCIe incl.)  identical=%s" % ("dh_bytes(k, encode(DH(m,G)))", LG, t1 - t0, ok), flush=True)
    assert ok
    t0 = time.time(); ex, est = eng.dh_exchange(k2, s, codec.pack_point((o.Gx, o.Gy))); t1 = time.time()
    ok = not est.any() and np.array_equal(ex, want)
    print("%-28s n=2^%d  GPU %.2fs (PCIe incl.)  identical=%s" % ("dh_exchange(k, m, G)", LG, t1 - t0, ok), flush=True)
    assert ok
    # the same multiplication from pinned host arrays and with constant-time table selection
    sp, pp, op_ = eng.host_array(s), eng.host_array(pts), eng.host_empty((n, 20))
    ref = oc.mul(oc.ENDO, s, pts)
    t0 = time.time(); got = eng.mul_endo(sp, pp, out=op_); t1 = time.time()
    ok = np.array_equal(got, ref)
    eng.host_timing(True)                            # the copies' durations are measured on request only (round 5): one more call for them
    op_[:] = 0
    got = eng.mul_endo(sp, pp, out=op_)
    eng.host_timing(False)
    st_ = eng.host_stats()
    ok = ok and np.array_equal(got, ref)
    print("%-28s n=2^%d  GPU %.3fs (PCIe incl., %d chunks, %.0f / %.0f GB/s)  identical=%s" % ("MUL_endo, pinned arrays", LG, t1 - t0, st_["chunks"],
          st_["gbs_h2d"], st_["gbs_d2h"], ok), flush=True)
    assert ok
    # round 5: the affine and the encoded I/O of MUL_* at this size (chunks of several generations: four elements share an inversion)
    aff_in = oc.r1_to_affine(pts)
    lifted = np.zeros((n, 20), dtype=np.uint64)
    lifted[:, 0:8] = aff_in; lifted[:, 8] = 1; lifted[:, 12:20] = aff_in
    want_aff = oc.r1_to_affine(oc.mul(oc.ENDO, s, lifted))
    ap, oap = eng.host_array(aff_in), eng.host_empty((n, 8))
    t0 = time.time(); got = eng.mul_affine(sp, ap, out=oap); t1 = time.time()
    ok = np.array_equal(got, want_aff)
    print("%-28s n=2^%d  GPU %.3fs (PCIe incl., %d chunks)  identical=%s" % ("MUL_endo affine I/O, pinned", LG, t1 - t0, eng.host_stats()["chunks"], ok), flush=True)
    assert ok
    t0 = time.time(); got, gst = eng.mul_bytes(s, oc.encode(aff_in)); t1 = time.time()
    ok = not gst.any() and np.array_equal(got, oc.encode(want_aff))
    print("%-28s n=2^%d  GPU %.3fs (PCIe incl., %d chunks)  identical=%s" % ("MUL_endo 32-byte I/O", LG, t1 - t0, eng.host_stats()["chunks"], ok), flush=True)
    assert ok
    eng.ct_select = True
    for name, gpu, cpu in (("MUL_endo, constant-time", lambda: eng.mul_endo(sp, pp, out=op_), lambda: ref),
                           ("MUL_windowed fixed, constant-time", lambda: eng.mul_windowed_fixed(s, tw), lambda: oc.mul(oc.WINDOWED, s, None, tw)),
                           ("comb keygen, constant-time", lambda: eng.comb_mul(s, comb)[0], lambda: b)):
        t0 = time.time(); x = gpu(); t1 = time.time()
        ok = np.array_equal(x, cpu())
        print("%-28s n=2^%d  GPU %.2fs (PCIe incl.)  identical=%s" % (name, LG, t1 - t0, ok), flush=True)
        assert ok
    # round 3: the mixed batch in constant-time mode (fused generations + the tail kernel with its riders), and both GPUs' worth of contexts behind one call
    t0 = time.time(); a = eng.mul_endo_mixed(s, pts, flags, te); t1 = time.time()
    want_mixed = np.where(flags.reshape(-1, 1) != 0, ref, oc.mul(oc.ENDO, s, None, te))
    print("%-28s n=2^%d  GPU %.2fs (PCIe incl.)  identical=%s" % ("mixed 50/50, constant-time", LG, t1 - t0, np.array_equal(a, want_mixed)), flush=True)
    assert np.array_equal(a, want_mixed)
from fourq_amd import MultiEngine
with MultiEngine([0, 0]) as multi:
    t0 = time.time(); a = multi.mul_endo(s, pts); t1 = time.time()
    print("%-28s n=2^%d  GPU %.2fs (PCIe incl.)  identical=%s" % ("MultiEngine([0, 0]).mul_endo", LG, t1 - t0, np.array_equal(a, ref)), flush=True)
    assert np.array_equal(a, ref)
    a, sa = multi.dh_exchange(k2, s, codec.pack_point((o.Gx, o.Gy)))
    ok = not sa.any() and np.array_equal(a, want)
    print("%-28s n=2^%d  identical=%s" % ("MultiEngine dh_exchange", LG, ok), flush=True)
    assert ok
print("SOAK OK")
