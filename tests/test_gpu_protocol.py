"""GPU parity of the protocol-level entry points and of the host-array pipeline:

  * fourq_dh_*_bytes_batch  -- decode -> DH -> encode on the device (draft-ladd-cfrg-4q.md:707-723;
    curve4q.py:49-96, :446-462, :41-46) against o.encode(*o.DH_endo(m, o.decode(B))), with every decode status of
    tests/golden/wire.json and the DH rejections of tests/golden/dh.json inside the batch;
  * fourq_dh_exchange_batch -- both halves chained on the device (curve4q.py:731);
  * the chunked H2D / kernel / D2H pipeline behind every host-array call: pageable and pinned callers, ragged
    multi-chunk batches, every route, compared with the C oracle;
  * GFp.select / GFp2.select on the device and DH_core with a caller-supplied `mul`.
"""
import random

import numpy as np
import pytest

import curve4q_oracle as o
import oracle_c as oc
from conftest import unhex
from fourq_amd import _lib, codec

pytestmark = pytest.mark.gpu

G = (o.Gx, o.Gy)
G1 = o.AffineToR1(o.Gx, o.Gy)


def seeded_scalars(seed, n):
    rng = random.Random(seed)
    return np.frombuffer(rng.getrandbits(256 * n).to_bytes(32 * n, "little"), dtype="<u8").reshape(n, 4).copy()


def oracle_dh_bytes(m, B, kind="endo", table=None):
    """(32 bytes, status) the way the reference's three steps compose; status as fourq_dh_*_bytes_batch reports it."""
    try:
        P = o.decode(bytes(B))
    except AttributeError:
        return bytes(32), _lib.BYTES_DECODE_BASE + _lib.DECODE_REF_ATTRIBUTE_ERROR
    except Exception as exc:
        code = _lib.DECODE_RESERVED_BIT if "reserved" in str(exc) else _lib.DECODE_NOT_ON_CURVE
        return bytes(32), _lib.BYTES_DECODE_BASE + code
    try:
        Q = (o.DH_endo if kind == "endo" else o.DH_windowed)(m, P, table=table)
    except Exception as exc:
        return bytes(32), (_lib.DH_NOT_ON_CURVE if "not on curve" in str(exc) else _lib.DH_NEUTRAL)
    return bytes(o.encode(*Q)), 0


@pytest.mark.parametrize("kind", ["endo", "windowed"])
def test_dh_bytes_golden_statuses(eng, golden, kind):
    w = golden("wire.json", raw=True)
    dh = golden("dh.json", raw=True)
    keys = [bytes.fromhex(r[1]) for r in w["roundtrip"][:24]]                      # valid public keys
    by_kind = {}
    for r in w["strings"]:                                                         # every decode verdict, a few each
        by_kind.setdefault(r[1] if r[1] == "ok" else r[2], []).append(bytes.fromhex(r[0]))
    assert len(by_kind) == 4
    for rows in by_kind.values():
        keys += rows[:6]
    small = [unhex(r[1]) for r in dh["reject"] if "neutral" in r[2] and unhex(r[1]) != G and o.PointOnCurve(unhex(r[1]))]
    keys += [bytes(o.encode(*P)) for P in small]                                   # on the curve, killed by the cofactor
    rng = random.Random(4242 + len(kind))
    ms = [rng.getrandbits(256) for _ in keys]
    ms[0], ms[1] = 0, o.N                                                          # neutral results from valid keys
    out, st = eng.dh_bytes(codec.pack_scalars(ms), np.frombuffer(b"".join(keys), dtype=np.uint8).reshape(-1, 32), kind=kind)
    want = [oracle_dh_bytes(m, B, kind) for m, B in zip(ms, keys)]
    assert [int(s) for s in st] == [s for _, s in want]
    assert [bytes(r) for r in out] == [b for b, _ in want]
    seen = set(int(s) for s in st)
    assert {0, _lib.DH_NEUTRAL, 16 + 1, 16 + 2, 16 + 3} <= seen


def test_dh_bytes_with_table_and_reference_shape(eng, golden):
    """table given: the reference still tests the decoded point but multiplies through the table (curve4q.py:209, :426)."""
    g392 = o.MUL_endo(392, G1)
    T = o.table_endo(g392)
    w = golden("wire.json", raw=True)
    keys = [bytes.fromhex(r[1]) for r in w["roundtrip"][:6]] + [bytes.fromhex(w["strings"][0][0])]
    ms = [random.Random(99).getrandbits(256) for _ in keys]
    out, st = eng.dh_bytes(codec.pack_scalars(ms), np.frombuffer(b"".join(keys), dtype=np.uint8).reshape(-1, 32), table=codec.pack_table(T))
    want = [oracle_dh_bytes(m, B, "endo", T) for m, B in zip(ms, keys)]
    assert [bytes(r) for r in out] == [b for b, _ in want] and [int(s) for s in st] == [s for _, s in want]


def test_dh_bytes_large_batch_both_parties(eng):
    """2^18 + 5 exchanges over the wire (split route, deferred normalisation, three pipeline chunks): both parties agree,
    a slice agrees with the oracle."""
    n = (1 << 18) + 5
    a, b = seeded_scalars(811, n), seeded_scalars(812, n)
    genc = np.repeat(eng.encode(codec.pack_point(G).reshape(1, 8)), n, axis=0)
    pa, s1 = eng.dh_bytes(a, genc)
    pb, s2 = eng.dh_bytes(b, genc)
    kab, s3 = eng.dh_bytes(a, pb)
    kba, s4 = eng.dh_bytes(b, pa)
    assert not (s1.any() or s2.any() or s3.any() or s4.any())
    assert np.array_equal(kab, kba)
    for i in (0, 1, n // 2, n - 1):
        m, bm = codec.unpack_scalars(a[i:i + 1])[0], codec.unpack_scalars(b[i:i + 1])[0]
        assert bytes(kab[i]) == bytes(o.encode(*o.DH_endo(m, o.DH_endo(bm, G))))
    # the same through device pointers, nothing but 32-byte strings crossing the ABI
    import torch
    dev = torch.device("cuda", 0)
    to_dev = lambda x, dt: torch.from_numpy(np.ascontiguousarray(x).view(dt)).to(dev)
    out_d = torch.empty((n, 32), dtype=torch.uint8, device=dev)
    st_d = torch.empty(n, dtype=torch.uint8, device=dev)
    eng.dh_bytes_dev(to_dev(a, np.int64), to_dev(pb, np.uint8), None, out_d, st_d, n)
    eng.sync()
    assert np.array_equal(out_d.cpu().numpy(), kab) and not bool(st_d.any())


def test_dh_exchange_on_device(eng, golden):
    raw = golden("dh.json", raw=True)
    for a, b, ab in unhex(raw["exchange"]):
        out, st = eng.dh_exchange(codec.pack_scalars([a]), codec.pack_scalars([b]), codec.pack_point(G))
        assert st[0] == 0 and codec.unpack_fp2s(out[0]) == ab
    n = 3000
    a, b = seeded_scalars(31, n), seeded_scalars(32, n)
    a[5] = 0                                  # second half ends in the neutral point
    b[9] = codec.pack_scalars([o.N])[0]       # first half does: its (zeroed) public key is then rejected as "not on curve"
    t392 = oc.table(oc.ENDO, codec.pack_point(o.MUL_endo(392, G1)))
    for table in (None, t392):
        out, st = eng.dh_exchange(a, b, codec.pack_point(G), table392=table)
        g = np.repeat(codec.pack_point(G).reshape(1, 8), n, axis=0)
        mid, s1 = oc.dh(oc.ENDO, b, g)
        want, s2 = oc.dh(oc.ENDO, a, mid)
        ws = np.where(s1 != 0, s1, s2)
        want[ws != 0] = 0
        assert np.array_equal(st, ws) and np.array_equal(out, want)
        assert st[5] == _lib.DH_NEUTRAL and st[9] == _lib.DH_NEUTRAL and not st[:5].any()
    bad = codec.pack_point(((1, 2), (3, 4)))  # base not on the curve: every exchange is rejected at the first step
    out, st = eng.dh_exchange(a[:8], b[:8], bad)
    assert (st == _lib.DH_NOT_ON_CURVE).all() and not out.any()
    # the same exchanges with the key-generation half through the comb of [392]G (fourq_dh_exchange_comb_batch): same outputs, same verdicts
    comb = eng.comb_table(codec.pack_point(o.MUL_endo(392, G1)))
    out_c, st_c = eng.dh_exchange_comb(a, b, comb)
    assert np.array_equal(st_c, ws) and np.array_equal(out_c, want)
    eng.comb_stage(comb)
    out_c, st_c = eng.dh_exchange_comb(a[:100], b[:100])            # None = the staged comb
    assert np.array_equal(st_c, ws[:100]) and np.array_equal(out_c, want[:100])
    m = 2 * eng.lanes + 321                                          # several pipeline chunks, the two-kernel route in the second half
    a2, b2 = seeded_scalars(33, m), seeded_scalars(34, m)
    out_c, st_c = eng.dh_exchange_comb(a2, b2)
    mid, s1 = oc.dh(oc.ENDO, b2, np.repeat(codec.pack_point(G).reshape(1, 8), m, axis=0))
    want2, s2 = oc.dh(oc.ENDO, a2, mid)
    assert not s1.any() and not s2.any() and not st_c.any() and np.array_equal(out_c, want2)


@pytest.mark.parametrize("pinned", [False, True])
def test_host_pipeline_chunks_and_pinned_memory(eng, pinned):
    """Ragged batches spanning several pipeline chunks, from pageable and from pinned arrays: every host-array entry
    point agrees with the C oracle (variable base) or with the single-chunk device path."""
    lanes = eng.lanes
    n = 2 * lanes + 1234
    put = (lambda x: eng.host_array(x)) if pinned else (lambda x: x)
    s = put(seeded_scalars(71, n))
    te = oc.table(oc.ENDO, codec.pack_point(G1))
    pts = put(eng.mul_endo_fixed(seeded_scalars(72, n), te))
    out_buf = eng.host_empty((n, 20)) if pinned else None
    want_mul = oc.mul(oc.ENDO, np.asarray(s), np.asarray(pts))
    for timing in (False, True):                               # copy durations only on request (fourq_ctx_set_host_timing): the events are not free
        eng.host_timing(timing)
        got = eng.mul_endo(s, pts, out=out_buf)
        st = eng.host_stats()
        assert st["chunks"] == 3 and st["pinned_in"] == int(pinned) and st["pinned_out"] == int(pinned)      # generation, generation, tail
        assert st["h2d_bytes"] == n * 192 and st["d2h_bytes"] == n * 160
        assert (st["h2d_ms"] > 0 and st["d2h_ms"] > 0) if timing else (st["h2d_ms"] == 0 and st["d2h_ms"] == 0)
        assert np.array_equal(got, want_mul)
    eng.host_timing(False)
    m = lanes + 77                                             # windowed: two chunks of the fused route
    assert np.array_equal(eng.mul_windowed(s[:m], pts[:m]), oc.mul(oc.WINDOWED, np.asarray(s[:m]), np.asarray(pts[:m])))
    tw = oc.table(oc.WINDOWED, codec.pack_point(G1))
    n4 = 4 * lanes + 9                                         # fixed base: a generation of the two-waves-per-SIMD ladders is 2 x lanes
    s4 = put(seeded_scalars(73, n4))
    got = eng.mul_windowed_fixed(s4, tw)                       # the result array is pageable either way: the larger uniform chunks of rounds 2-4
    assert eng.host_stats()["chunks"] == 2                     # 4 x lanes + tail
    want4 = oc.mul(oc.WINDOWED, np.asarray(s4), None, tw)
    assert np.array_equal(got, want4)
    if pinned:                                                 # every array pinned: generation, generation, tail
        o4 = eng.host_empty((n4, 20))
        assert np.array_equal(eng.mul_windowed_fixed(s4, tw, out=o4), want4) and eng.host_stats()["chunks"] == 3
        eng.host_free(o4)
    g = np.repeat(codec.pack_point(G).reshape(1, 8), n, axis=0)
    aff, st0 = oc.dh(oc.ENDO, seeded_scalars(72, n), g)        # affine inputs: DH_endo(k_i, G)
    assert not st0.any()
    aff = put(aff)
    out, st = eng.dh_endo(s, aff)                              # n >= 2 x lanes: split route, chunk = 4 x lanes -> one chunk
    want, ws = oc.dh(oc.ENDO, np.asarray(s), np.asarray(aff))
    assert np.array_equal(out, want) and np.array_equal(st, ws)
    flags = put((np.frombuffer(random.Random(74).getrandbits(8 * n4).to_bytes(n4, "little"), dtype=np.uint8) & 1).copy())
    p4 = put(eng.mul_endo_fixed(seeded_scalars(75, n4), te))
    got = eng.mul_endo_mixed(s4, p4, flags, te)
    want = np.where(np.asarray(flags).reshape(-1, 1) != 0, oc.mul(oc.ENDO, np.asarray(s4), np.asarray(p4)), oc.mul(oc.ENDO, np.asarray(s4), None, te))
    assert np.array_equal(got, want)
    for arr in (s, pts, s4, aff, flags, p4) + ((out_buf,) if pinned else ()):
        if pinned:
            eng.host_free(arr)


@pytest.mark.parametrize("slots,gens,host_wait", [(4, 0, 0), (4, 2, 0), (2, 1, 0), (3, 4, 0), (6, 3, 0), (4, 2, 1), (3, 0, 1)])
def test_host_pipeline_shapes_slots_handed_on_by_the_gpu(slots, gens, host_wait):
    """Round 5: the chunks of a large host-array call are one generation first and last and as large in between as the copies' lead allows
    (fourq_amd/csrc/pipeline_plan.h; `gens` > 0 forces [one] [`gens`]* [rest] [one] [tail] instead), cycled through
    `slots` device slots that the GPU hands on itself (the copy-in stream waits for the event behind the slot's previous copy-out; the host
    never blocks while it enqueues).  Every shape of that plan -- fewer chunks than slots, slots reused several times, the host-side
    hand-over of rounds 2-4, pageable arrays through the bounce slots -- must give the C oracle's words: 7 generations + a ragged tail."""
    import os
    from fourq_amd import Engine
    knobs = {"FOURQ_PIPE_SLOTS": str(slots), "FOURQ_PIPE_GENS": str(gens), "FOURQ_PIPE_HOST_WAIT": str(host_wait)}
    saved = {k: os.environ.get(k) for k in knobs}
    os.environ.update(knobs)
    try:
        e = Engine(0)
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    try:
        lanes = e.lanes
        n = 7 * lanes + 555
        te = oc.table(oc.ENDO, codec.pack_point(G1))
        s_h = seeded_scalars(171, n)
        p_h = e.mul_endo_fixed(seeded_scalars(172, n), te)
        want = oc.mul(oc.ENDO, s_h, p_h)
        inner = 5                                                  # generations between the first and the last whole one
        pieces = 1 + (inner + gens - 1) // gens + 1 + 1 if gens else 8         # planned: raw R1 I/O stays at one generation per chunk (+ the tail)
        s_p, p_p, o_p = e.host_array(s_h), e.host_array(p_h), e.host_empty((n, 20))
        for rep in range(2):                                       # the second call finds the slots' events already used
            got = e.mul_endo(s_p, p_p, out=o_p)
            st = e.host_stats()
            # planned chunks follow the context's own measurements from the second call on (round 6): in constant-time mode the slower kernels
            # leave room for a two-generation chunk or two
            assert (st["chunks"] == pieces if gens or rep == 0 else pieces - 2 <= st["chunks"] <= pieces) and st["pinned_in"] == 1 and st["pinned_out"] == 1
            assert st["planned_from_measurement"] == rep and st["measured_kernel_ns_per_elem"] > 0
            assert np.array_equal(got, want), "pinned, call %d" % rep
            o_p[:] = 0
        assert np.array_equal(e.mul_endo(s_h, p_h), want), "pageable"
        assert e.host_stats()["pinned_in"] == 0
        # a format with several kernels per chunk and two output arrays
        a_p, oa_p = e.host_array(oc.r1_to_affine(p_h)), e.host_empty((n, 8))      # the same points, affine
        got = e.mul_affine(s_p, a_p, out=oa_p)
        assert np.array_equal(got, oc.r1_to_affine(want))
        for arr in (s_p, p_p, o_p, a_p, oa_p):
            e.host_free(arr)
    finally:
        e.close()


def test_planner_inputs_are_measured_by_the_context_itself():
    """the chunk planner's inputs are no longer constants of one box and one mode.  A context's first multi-chunk call of a
    route plans with the compiled-in guess (x 1.3 in constant-time mode); every such call times one middle chunk -- its kernels and its
    copies, both directions of the link busy -- and the next call of that route on that context is planned with those figures.  A
    constant-time context therefore plans for ITS kernels; the figures are per route and per selection mode."""
    from fourq_amd import Engine
    with Engine(0) as e:
        lanes = e.lanes
        n = 8 * lanes
        te = oc.table(oc.ENDO, codec.pack_point(G1))
        s_h = seeded_scalars(181, n)
        p_h = e.mul_endo_fixed(seeded_scalars(182, n), te)
        s_p, p_p, o_p = e.host_array(s_h), e.host_array(p_h), e.host_empty((n, 20))
        seen = {}
        for ct in (False, True):
            e.ct_select = ct
            e.mul_endo(s_p, p_p, out=o_p)
            st = e.host_stats()
            assert st["planned_from_measurement"] == 0 and abs(st["planned_kernel_ns_per_elem"] - 4.63 * (1.3 if ct else 1.0)) < 1e-9
            if not ct:
                assert st["planned_link_in_gbs"] == 48.0 and st["planned_link_out_gbs"] == 48.0          # nothing measured yet
            first = st["measured_kernel_ns_per_elem"]
            assert 3.5 < first < 9.0
            e.mul_endo(s_p, p_p, out=o_p)
            st = e.host_stats()
            assert st["planned_from_measurement"] == 1 and st["planned_kernel_ns_per_elem"] == first
            assert 25.0 < st["planned_link_in_gbs"] < 70.0 and 25.0 < st["planned_link_out_gbs"] < 70.0, st
            seen[ct] = st["measured_kernel_ns_per_elem"]
        assert seen[True] > 1.08 * seen[False], seen                     # the constant-time ladder reads the whole table at every step
        want = oc.mul(oc.ENDO, s_h, p_h)
        assert np.array_equal(o_p, want)
        # another route on the same context starts from its own guess again
        a_p, oa_p = e.host_array(oc.r1_to_affine(p_h)), e.host_empty((n, 8))
        e.ct_select = False
        e.mul_affine(s_p, a_p, out=oa_p)
        st = e.host_stats()
        assert st["planned_from_measurement"] == 0 and st["planned_link_in_gbs"] != 48.0                 # the link's rates are the context's, not the route's
        assert np.array_equal(oa_p, oc.r1_to_affine(want))
        # the per-chunk stamps of a call made under host_timing (fourq_ctx_host_chunk_stamps): one row per chunk, each stage in order
        e.host_timing(True)
        e.mul_endo(s_p, p_p, out=o_p)
        rows = e.host_chunk_stamps()
        e.host_timing(False)
        assert len(rows) == e.host_stats()["chunks"]
        for k, (i0, i1, o0, o1, k0, k1) in enumerate(rows):
            assert i0 <= i1 <= k0 + 0.05 and k0 < k1 <= o0 + 0.05 and o0 < o1, (k, rows[k])
        assert all(rows[k][4] >= rows[k - 1][5] - 0.05 for k in range(1, len(rows)))                     # one kernel stream: chunk k's kernels behind chunk k-1's
        for arr in (s_p, p_p, o_p, a_p, oa_p):
            e.host_free(arr)


def test_fused_io_flags_give_the_three_kernel_route_s_words():
    """Round 6: on whole fused generations the affine / encoded flavours of MUL_* hand the ladder affine rows and take (X, Y, Z) rows back
    (LadderArgs::io) -- no lift kernel, no R1 rows.  Same words as round 5's route (FOURQ_FUSED_IO=0, a test hook) and as the C oracle,
    host arrays (chunks of whole generations + a tail on the two-lane kernels) and one device-resident launch, both ladders, both modes."""
    import os
    import torch
    from fourq_amd import Engine
    dev = torch.device("cuda", 0)
    outs = {}
    for hook in ("1", "0"):
        saved = {k: os.environ.get(k) for k in ("FOURQ_FUSED_IO", "FOURQ_DEBUG_ROUTES")}
        os.environ.update(FOURQ_FUSED_IO=hook, FOURQ_DEBUG_ROUTES="1")
        try:
            e = Engine(0)
        finally:
            for k, v in saved.items():
                os.environ.pop(k, None) if v is None else os.environ.__setitem__(k, v)
        try:
            lanes = e.lanes
            n = 3 * lanes + 77
            te = oc.table(oc.ENDO, codec.pack_point(G1))
            s_h = seeded_scalars(191, n)
            r1_h = e.mul_endo_fixed(seeded_scalars(192, n), te)
            aff = oc.r1_to_affine(r1_h)
            enc = oc.encode(aff)
            enc[5, 15] |= 0x80                                            # one key that does not decode: reserved bit
            for ct in (False, True):
                e.ct_select = ct
                for kind, okind in (("endo", oc.ENDO), ("windowed", oc.WINDOWED)):
                    got_a = e.mul_affine(s_h, aff, kind=kind)
                    got_b, st_b = e.mul_bytes(s_h, enc, kind=kind)
                    m = 2 * lanes                                         # device-resident, whole generations: one launch with the flags
                    sd, ad = (torch.from_numpy(np.ascontiguousarray(a[:m]).view(np.int64)).to(dev) for a in (s_h, aff))
                    od = torch.empty((m, 8), dtype=torch.int64, device=dev)
                    e.mul_affine_dev(sd, ad, od, m, kind=kind)
                    e.sync()
                    assert np.array_equal(od.cpu().numpy().view(np.uint64), got_a[:m])
                    outs[(hook, ct, kind)] = (got_a, got_b, st_b)
                    if hook == "1":
                        lifted = np.zeros((n, 20), dtype=np.uint64)
                        lifted[:, 0:8] = aff; lifted[:, 8] = 1; lifted[:, 12:20] = aff
                        want = oc.r1_to_affine(oc.mul(okind, s_h, lifted))
                        assert np.array_equal(got_a, want), (ct, kind)
                        assert st_b[5] == 16 + 1 and not got_b[5].any() and not np.delete(st_b, 5).any()
                        assert np.array_equal(np.delete(got_b, 5, axis=0), np.delete(oc.encode(want), 5, axis=0)), (ct, kind)
        finally:
            e.close()
    for ct in (False, True):
        for kind in ("endo", "windowed"):
            a, b = outs[("1", ct, kind)], outs[("0", ct, kind)]
            assert all(np.array_equal(x, y) for x, y in zip(a, b)), (ct, kind)


@pytest.mark.parametrize("pinned", [False, True])
def test_small_host_call_runs_in_order_on_the_context_stream(eng, pinned):
    """A call of one chunk runs in order on the context's stream instead of the three-stream pipeline, and at most 64 KiB (the
    reference-shaped call is a batch of one) the kernels read and write pinned host memory in place: same results, no device copies
    hence no byte counts (they used to be reported for what are CPU memcpys).  Copy durations are measured on request only
    (round 5: fourq_ctx_set_host_timing) and never for the in-place calls."""
    put = (lambda x: eng.host_array(x)) if pinned else (lambda x: x)
    te = oc.table(oc.ENDO, codec.pack_point(G1))
    for n in (1, 7, 185, 186, 2977, 2978):                     # 185 x (32 + 160 + 160) B in 256-byte-aligned arrays: the last size the kernels read and write in pinned host memory directly (64 KiB); 2977: just under 1 MiB; 2978: just over
        s = put(seeded_scalars(91, n))
        pts = put(eng.mul_endo_fixed(seeded_scalars(92, n), te))
        got = eng.mul_endo(s, pts)
        st = eng.host_stats()
        assert np.array_equal(got, oc.mul(oc.ENDO, np.asarray(s), np.asarray(pts)))
        in_place = n <= 185
        assert st["chunks"] == 1 and st["pinned_in"] == int(pinned)
        assert (st["h2d_bytes"], st["d2h_bytes"]) == ((0, 0) if in_place else (n * 192, n * 160))
        assert st["h2d_ms"] == 0 and st["d2h_ms"] == 0
        eng.host_timing(True)
        try:
            assert np.array_equal(eng.mul_endo(s, pts), got)
            st = eng.host_stats()
            assert (st["h2d_ms"] == 0 and st["d2h_ms"] == 0) if in_place else (st["h2d_ms"] > 0 and st["d2h_ms"] > 0)
        finally:
            eng.host_timing(False)
        g = put(np.repeat(codec.pack_point(G).reshape(1, 8), n, axis=0))
        out, status = eng.dh_endo(s, g)
        want, ws = oc.dh(oc.ENDO, np.asarray(s), np.asarray(g))
        assert np.array_equal(out, want) and np.array_equal(status, ws)
        if pinned:
            for arr in (s, pts, g):
                eng.host_free(arr)


def test_set_stream_restages_the_fixed_base_table(eng):
    import torch
    tw = oc.table(oc.WINDOWED, codec.pack_point(G1))
    te = oc.table(oc.ENDO, codec.pack_point(G1))
    n = 5000
    s = seeded_scalars(81, n)
    want_w = oc.mul(oc.WINDOWED, s, None, tw)
    want_e = oc.mul(oc.ENDO, s, None, te)
    dev = torch.device("cuda", 0)
    s_d = torch.from_numpy(s.view(np.int64)).to(dev)
    out = torch.empty((n, 20), dtype=torch.int64, device=dev)
    other = torch.cuda.Stream(device=dev)
    for stream, table, want in ((None, tw, want_w), (other, tw, want_w), (None, te, want_e), (other, te, want_e), (other, tw, want_w)):
        eng.set_stream(stream.cuda_stream if stream is not None else None)
        eng.mul_windowed_fixed_dev(s_d, table, out, n) if table is tw else eng.mul_endo_fixed_dev(s_d, table, out, n)
        eng.sync()
        assert np.array_equal(out.cpu().numpy().view(np.uint64), want)
    eng.set_stream(None)


def test_select_and_dh_core_with_any_callable(golden):
    from fourq_amd import curve4q as c
    rng = random.Random(5)
    for _ in range(16):
        x, y = rng.getrandbits(127), rng.getrandbits(127)
        a, b = (rng.getrandbits(127), rng.getrandbits(127)), (rng.getrandbits(127), rng.getrandbits(127))
        for bit in (0, 1):
            assert c.GFp.select(bit, x, y) == o.fp_select(bit, x, y) == (x if bit else y)
            assert c.GFp2.select(bit, a, b) == o.f2_select(bit, a, b) == (a if bit else b)
    assert c.GFp.select(1, (1 << 128) - 1, 0) == (1 << 128) - 1            # raw bits, not reduced (as the reference)
    raw = golden("dh.json", raw=True)
    calls = []

    def my_mul(m, Q, table=None):              # a caller's own multiplication routine, as DH_core allows (curve4q.py:446)
        calls.append(m)
        return c.MUL_windowed(m, Q, table=table)

    for m, Pt, e, w in unhex(raw["dh"])[:3]:
        assert c.DH_core(m, Pt, my_mul) == w == o.DH_core(m, Pt, o.MUL_windowed)
    assert len(calls) == 3
    for m, Pt, msg in raw["reject"][:2]:
        with pytest.raises(Exception) as ei:
            c.DH_core(int(m, 16), unhex(Pt), my_mul)
        assert str(ei.value) == msg


def test_protocol_vectors_of_the_reference(eng, golden):
    """tests/golden/protocol.json was produced by the REFERENCE composing its own decode, DH_* and encode: the one-call
    device path must give the same 32 bytes, or the status of the same failure."""
    g = golden("protocol.json", raw=True)
    rows = g["dh_bytes"]
    s = codec.pack_scalars([int(r[0], 16) for r in rows])
    keys = np.frombuffer(b"".join(bytes.fromhex(r[1]) for r in rows), dtype=np.uint8).reshape(-1, 32)
    status_of = {"Malformed point: reserved bit is not zero": 16 + _lib.DECODE_RESERVED_BIT, "Point not on curve": 16 + _lib.DECODE_NOT_ON_CURVE,
                 "type object 'GFp' has no attribute 'two'": 16 + _lib.DECODE_REF_ATTRIBUTE_ERROR,
                 "DH computation resulted in neutral point": _lib.DH_NEUTRAL}
    for kind, col in (("endo", 2), ("windowed", 3)):
        out, st = eng.dh_bytes(s, keys, kind=kind)
        for r, got, code in zip(rows, out, st):
            if r[col][0] == "ok":
                assert code == 0 and bytes(got).hex() == r[col][1]
            else:
                assert code == status_of[r[col][1]] and not got.any()
    assert len({r[2][1] for r in rows if r[2][0] != "ok"}) == 4          # all four kinds of failure are in the fixture


def test_select_vectors_of_the_reference(golden):
    from fourq_amd import curve4q as c
    g = golden("protocol.json", raw=True)
    for bit, x, y, r in unhex(g["select"])[:16]:
        assert c.GFp.select(bit, x, y) == r
    for bit, a, b, r in unhex(g["select2"])[:16]:
        assert c.GFp2.select(bit, a, b) == r
