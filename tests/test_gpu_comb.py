"""Fixed-base comb: affine [m]B from a 1 024-point table (w = 9, v = 4, e = 7, d = 28; the constant-time mode scans
an 80-point comb, w = v = 5, kept in the same table object).  Parity is at the affine level
(the draft's "MAY use any method ... provided that it agrees", draft-ladd-cfrg-4q.md:725-729): outputs must
equal R1toAffine(MUL_endo(m, B)) and, for B = [392]G, DH_endo(m, G)."""
import random

import numpy as np
import pytest

import curve4q_oracle as o
import oracle_c as oc
from fourq_amd import codec

pytestmark = pytest.mark.gpu

G1 = o.AffineToR1(o.Gx, o.Gy)


def seeded_scalars(seed, n):
    rng = random.Random(seed)
    return np.frombuffer(rng.getrandbits(256 * n).to_bytes(32 * n, "little"), dtype="<u8").reshape(n, 4).copy()


def test_comb_table_entries(eng):
    from fourq_amd import _lib
    comb = eng.comb_table(codec.pack_point(G1)).reshape(_lib.COMB_POINTS, 12)
    assert _lib.COMB_POINTS == 1024 + 80
    for t in (0, 1, 255, 256, 401, 767, 1023, 1024, 1025, 1040, 1068, 1103):       # both shapes of the table object
        (W, V, E, D), s = ((9, 4, 7, 28), t) if t < 1024 else ((5, 5, 10, 50), t - 1024)
        j, u = s >> (W - 1), s & ((1 << (W - 1)) - 1)
        m = (1 << (E * j)) * (1 + sum(((u >> r) & 1) << (D * (r + 1)) for r in range(W - 1)))
        x, y = o.R1toAffine(o.MUL_endo(m % o.N, G1))
        want = (o.f2_add(x, y), o.f2_sub(y, x), o.f2_mul(o.TWO_D, o.f2_mul(x, y)))
        assert codec.unpack_fp2s(comb[t]) == want, t


def test_comb_equals_mul_endo_affine(eng):
    rng = random.Random(77)
    B = o.MUL_endo(rng.getrandbits(256), G1)                    # a projective base point of order N
    comb = eng.comb_table(codec.pack_point(B))
    ms = [0, 1, 2, 3, o.N - 1, o.N, o.N + 1, 2 * o.N, (1 << 256) - 1, 1 << 255, 1 << 28, (1 << 28) - 1, 1 << 36, (1 << 36) - 1, 1 << 50, (1 << 50) - 1, (1 << 252) - 1, 1 << 216, (1 << 245) + 1] + [rng.getrandbits(256) for _ in range(40)]
    out, st = eng.comb_mul(codec.pack_scalars(ms), comb)
    for m, got, s in zip(ms, out, st):
        want = o.R1toAffine(o.MUL_endo(m, B))
        if want == (o.Ox, o.Oy):
            assert s == 2 and not got.any()
        else:
            assert s == 0 and codec.unpack_fp2s(got) == want, hex(m)


def test_comb_keygen_equals_dh_fixed_base_full_batch(eng):
    """2^18 key generations: comb([392]G) against DH_endo(m, G, table_endo([392]G)) (curve4q.py:743-762) on the GPU,
    and a 2^12 slice against the C oracle."""
    n = 1 << 18
    s = seeded_scalars(8181, n)
    g392 = codec.pack_point(o.MUL_endo(392, G1))
    comb, te = eng.comb_table(g392), eng.table_endo(g392)
    got, st = eng.comb_mul(s, comb)
    gaff = np.repeat(codec.pack_point((o.Gx, o.Gy)).reshape(1, 8), n, axis=0)
    ref, rst = eng.dh_endo(s, gaff, te)
    assert not st.any() and not rst.any() and np.array_equal(got, ref)
    k = 1 << 12
    want, wst = oc.dh(oc.ENDO, s[:k], gaff[:k])
    assert not wst.any() and np.array_equal(got[:k], want)


@pytest.mark.parametrize("n", [1, 63, 65, 255, 257, 16385, 40000, 70001])      # block widths 64, 128, 256, 512 (1 024: the 2^18 test), ragged tails
def test_comb_ragged(eng, n):
    s = seeded_scalars(900 + n, n)
    comb = eng.comb_table(codec.pack_point(G1))
    got, st = eng.comb_mul(s, comb)
    te = oc.table(oc.ENDO, codec.pack_point(G1))
    full = oc.mul(oc.ENDO, s, None, te)
    want = eng.prim("PT_R1TOAFFINE", full)
    assert not st.any() and np.array_equal(got, want)


@pytest.mark.parametrize("quad", ["four, two, one", "two, one", "one lane per element"])
def test_small_comb_batches_four_lanes_per_element_and_one(quad, monkeypatch):
    """Key generation for at most a quarter generation runs four lanes per element, up to half a generation two, with the entries
    gathered from the table in memory (comb_quad_kernel<CT, 4 | 2>); FOURQ_QUAD_MAX=0 keeps the four-lane form out, FOURQ_PAIR_MAX=0
    leaves the one-lane kernel with the table staged in LDS.  Same affine points and verdicts every way: edge scalars (0, N, 2N are the
    neutral point), ragged sizes around the blocks and the route switches, both selection modes, against the reference-shaped computation."""
    from fourq_amd import Engine
    monkeypatch.delenv("FOURQ_QUAD_MAX", raising=False)
    monkeypatch.delenv("FOURQ_PAIR_MAX", raising=False)
    if quad == "two, one":
        monkeypatch.setenv("FOURQ_QUAD_MAX", "0")
    elif quad == "one lane per element":
        monkeypatch.setenv("FOURQ_PAIR_MAX", "0")
    with Engine(0) as e:
        comb = e.comb_table(codec.pack_point(G1))
        te = oc.table(oc.ENDO, codec.pack_point(G1))
        N = o.N
        edge = [0, 1, 2, 3, N - 1, N, N + 1, 2 * N, 2 * N + 1, 1 << 255, (1 << 256) - 1, 1 << 28, (1 << 28) - 1, (1 << 50) - 1, 1 << 50, (1 << 252) - 1]
        big = e.lanes // 2 + 1
        s = seeded_scalars(4711, big)
        want = e.prim("PT_R1TOAFFINE", oc.mul(oc.ENDO, s, None, te))
        for ct in (False, True):                                  # the constant-time shape (80 points, whole blocks read) has its own quad kernel
            e.ct_select = ct
            out, st = e.comb_mul(codec.pack_scalars(edge), comb)
            for m, got, v in zip(edge, out, st):
                ref = o.R1toAffine(o.MUL_endo(m, G1))
                if ref == (o.Ox, o.Oy):
                    assert v == 2 and not got.any(), (ct, hex(m))
                else:
                    assert v == 0 and codec.unpack_fp2s(got) == ref, (ct, hex(m))
            for m in (1, 2, 63, 64, 65, 127, 128, 129, 4097, e.lanes // 4, e.lanes // 4 + 1, e.lanes // 2, big):
                got, v = e.comb_mul(s[:m], comb)
                assert not v.any() and np.array_equal(got, want[:m]), (quad, ct, m)


def test_comb_keygen_then_dh_in_a_hip_graph(eng):
    """Key generation through the comb followed by the peer's variable-base DH on the keys, captured into one HIP graph and
    replayed: the comb's 144 KB of dynamic LDS is granted at context creation, so the launch itself is a pure enqueue."""
    import torch
    dev = torch.device("cuda", 0)
    n = 5000
    b_h, a_h = seeded_scalars(71, n), seeded_scalars(72, n)
    comb = eng.comb_table(codec.pack_point(o.MUL_endo(392, G1)))
    b, a = (torch.from_numpy(x.view(np.int64)).to(dev) for x in (b_h, a_h))
    keys = torch.empty((n, 8), dtype=torch.int64, device=dev)
    shared = torch.empty((n, 8), dtype=torch.int64, device=dev)
    st1 = torch.empty(n, dtype=torch.uint8, device=dev)
    st2 = torch.empty(n, dtype=torch.uint8, device=dev)
    side = torch.cuda.Stream(device=dev)
    eng.set_stream(side.cuda_stream)
    try:
        eng.comb_mul_dev(b, comb, keys, st1, n)            # stages the comb and sizes the context's buffers outside the capture
        eng.dh_endo_dev(a, keys, None, shared, st2, n)
        eng.sync()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.stream(side):
            graph.capture_begin()
            eng.comb_mul_dev(b, comb, keys, st1, n)
            eng.dh_endo_dev(a, keys, None, shared, st2, n)
            graph.capture_end()
        keys.zero_(); shared.zero_()
        graph.replay()
        torch.cuda.synchronize()
    finally:
        eng.set_stream(None)
    gaff = np.repeat(codec.pack_point((o.Gx, o.Gy)).reshape(1, 8), n, axis=0)
    want_keys, wst = oc.dh(oc.ENDO, b_h, gaff)
    assert not wst.any() and not st1.cpu().numpy().any() and np.array_equal(keys.cpu().numpy().view(np.uint64), want_keys)
    want_shared, wst2 = oc.dh(oc.ENDO, a_h, want_keys)
    assert not wst2.any() and not st2.cpu().numpy().any() and np.array_equal(shared.cpu().numpy().view(np.uint64), want_shared)


def test_reserved_context_is_capturable_without_a_warm_up_call(monkeypatch):
    """fourq_ctx_reserve + fourq_comb_stage: a fresh context whose very first DH-sized calls are issued INSIDE a graph capture.
    FOURQ_NORM_K=2 makes every DH batch take the deferred-normalisation route, whose planes would otherwise be allocated (and
    the stream synchronised) by the first call that needs them -- which a capture does not allow."""
    import torch
    from fourq_amd import Engine, FourQError
    monkeypatch.setenv("FOURQ_NORM_K", "2")
    dev = torch.device("cuda", 0)
    n = 3001
    b_h, a_h = seeded_scalars(81, n), seeded_scalars(82, n)
    side = torch.cuda.Stream(device=dev)
    with Engine(0, stream=side.cuda_stream) as e:
        b, a = (torch.from_numpy(x.view(np.int64)).to(dev) for x in (b_h, a_h))
        keys = torch.zeros((n, 8), dtype=torch.int64, device=dev)
        shared = torch.zeros((n, 8), dtype=torch.int64, device=dev)
        st1 = torch.empty(n, dtype=torch.uint8, device=dev)
        st2 = torch.empty(n, dtype=torch.uint8, device=dev)
        with pytest.raises(FourQError):
            e.comb_mul_dev(b, None, keys, st1, n)          # nothing staged yet: NULL table is an error, not a crash
        comb = e.comb_table(codec.pack_point(o.MUL_endo(392, G1)))
        e.comb_stage(comb)
        e.reserve(n)
        e.sync()
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.stream(side):
            graph.capture_begin()
            e.comb_mul_dev(b, None, keys, st1, n)          # None = the staged comb: no host-side compare, no upload
            e.dh_endo_dev(a, keys, None, shared, st2, n)
            graph.capture_end()
        graph.replay()
        torch.cuda.synchronize()
        gaff = np.repeat(codec.pack_point((o.Gx, o.Gy)).reshape(1, 8), n, axis=0)
        want_keys, _ = oc.dh(oc.ENDO, b_h, gaff)
        want_shared, _ = oc.dh(oc.ENDO, a_h, want_keys)
        assert np.array_equal(keys.cpu().numpy().view(np.uint64), want_keys)
        assert np.array_equal(shared.cpu().numpy().view(np.uint64), want_shared)
        assert not st1.cpu().numpy().any() and not st2.cpu().numpy().any()
        # a change of stream drops the device copy of the staged comb, not the table: None still means "the table given to comb_stage"
        other = torch.cuda.Stream(device=dev)
        e.set_stream(other.cuda_stream)
        keys.zero_()
        torch.cuda.synchronize()
        e.comb_mul_dev(b, None, keys, st1, n)
        e.sync()
        assert np.array_equal(keys.cpu().numpy().view(np.uint64), want_keys)


def test_staging_a_table_inside_a_capture_is_refused_with_a_message():
    """a table that has to be uploaded must be staged outside a stream capture (the upload would be captured reading the
    context's mutable host copy, and the event that guards that copy would become a captured event).  The call says so and the
    context stays usable."""
    import torch
    from fourq_amd import Engine, FourQError
    dev = torch.device("cuda", 0)
    n = 2048
    s_h = seeded_scalars(91, n)
    side = torch.cuda.Stream(device=dev)
    with Engine(0, stream=side.cuda_stream) as e:
        comb = e.comb_table(codec.pack_point(o.MUL_endo(392, G1)))
        te = e.table_endo(codec.pack_point(G1))
        s = torch.from_numpy(s_h.view(np.int64)).to(dev)
        keys = torch.zeros((n, 8), dtype=torch.int64, device=dev)
        out = torch.zeros((n, 20), dtype=torch.int64, device=dev)
        st = torch.empty(n, dtype=torch.uint8, device=dev)
        e.sync()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.stream(side):
            graph.capture_begin()
            try:
                with pytest.raises(FourQError, match="staged while the stream is being captured"):
                    e.comb_mul_dev(s, comb, keys, st, n)                   # nothing staged yet
                with pytest.raises(FourQError, match="staged while the stream is being captured"):
                    e.mul_endo_fixed_dev(s, te, out, n)
            finally:
                graph.capture_end()
        e.comb_mul_dev(s, comb, keys, st, n)                               # outside: stages and runs
        e.sync()
        gaff = np.repeat(codec.pack_point((o.Gx, o.Gy)).reshape(1, 8), n, axis=0)
        want, wst = oc.dh(oc.ENDO, s_h, gaff)
        assert not wst.any() and np.array_equal(keys.cpu().numpy().view(np.uint64), want)
