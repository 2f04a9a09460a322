"""GPU parity of the scalar-multiplication hot path (MUL_endo / MUL_windowed, variable and fixed
base) through the C ABI: golden vectors, the reference's KAT chain, full-batch comparison against
the C oracle, ragged/empty batches, and size-independent properties at BASELINE.json sizes."""
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


def torsion_points(eng, seed, n):
    """n projective N-torsion points: raw R1 outputs of fixed-base [k_i]G."""
    tbl = oc.table(oc.ENDO, codec.pack_point(G1))
    return eng.mul_endo_fixed(seeded_scalars(seed, n), tbl)


def test_tables(eng, golden):
    for Pt, tw, te in golden("tables.json")["tables"]:
        assert codec.unpack_table(eng.table_windowed(codec.pack_point(Pt))) == list(tw)
        assert codec.unpack_table(eng.table_endo(codec.pack_point(Pt))) == list(te)


def test_mul_golden(eng, golden):
    g = golden("mul.json")
    rows = g["var"] + g["edge"]
    s = codec.pack_scalars([r[0] for r in rows])
    p = codec.pack_points([r[1] for r in rows], 5)
    assert codec.unpack_points(eng.mul_endo(s, p)) == [r[2] for r in rows]
    assert codec.unpack_points(eng.mul_windowed(s, p)) == [r[3] for r in rows]
    for blk in g["fixed"]:
        s = codec.pack_scalars([r[0] for r in blk["rows"]])
        assert codec.unpack_points(eng.mul_endo_fixed(s, codec.pack_table(blk["table_endo"]))) == [r[1] for r in blk["rows"]]
        assert codec.unpack_points(eng.mul_windowed_fixed(s, codec.pack_table(blk["table_windowed"]))) == [r[2] for r in blk["rows"]]


@pytest.mark.parametrize("kind", ["endo", "windowed"])
def test_mul_chain_kat(eng, golden, kind):
    """curve4q.py:549-567: 1000 chained scalar mults, outputs fed back as projective inputs."""
    from conftest import unhex
    fn = eng.mul_endo if kind == "endo" else eng.mul_windowed
    A = codec.pack_point(G1).reshape(1, 20)
    for m in o.kat_scalars(1000):
        A = fn(codec.pack_scalars([m]), A)
    assert o.R1toAffine(codec.unpack_fp2s(A[0])) == unhex(golden("kat.json", raw=True)["mulP"])


def test_mul_one_two_and_fixed_equals_variable(eng):   # curve4q.py:571-598, :677-704
    g = codec.pack_point(G1).reshape(1, 20)
    dbl = o.R1toAffine(o.DBL(G1))
    rng = random.Random(5)
    ms = [rng.getrandbits(256) for _ in range(10)]
    for var, fixed, tab in ((eng.mul_endo, eng.mul_endo_fixed, eng.table_endo), (eng.mul_windowed, eng.mul_windowed_fixed, eng.table_windowed)):
        T = tab(g)
        for m, want in ((1, (o.Gx, o.Gy)), (2, dbl)):
            s = codec.pack_scalars([m])
            assert o.R1toAffine(codec.unpack_fp2s(var(s, g)[0])) == want
            assert o.R1toAffine(codec.unpack_fp2s(fixed(s, T)[0])) == want
        s = codec.pack_scalars(ms)
        assert np.array_equal(fixed(s, T), var(s, np.repeat(g, len(ms), axis=0)))   # full R1 tuples, as :593, :699


@pytest.mark.parametrize("n", [0, 1, 2, 63, 64, 65, 255, 257, 1000])
def test_ragged_batches(eng, n):
    s = seeded_scalars(77 + n, n)
    pts = torsion_points(eng, 1234 + n, n) if n else np.empty((0, 20), dtype=np.uint64)
    assert np.array_equal(eng.mul_endo(s, pts), oc.mul(oc.ENDO, s, pts) if n else np.empty((0, 20), dtype=np.uint64))
    assert np.array_equal(eng.mul_windowed(s, pts), oc.mul(oc.WINDOWED, s, pts) if n else np.empty((0, 20), dtype=np.uint64))
    tbl = oc.table(oc.ENDO, codec.pack_point(G1))
    assert np.array_equal(eng.mul_endo_fixed(s, tbl), oc.mul(oc.ENDO, s, None, tbl) if n else np.empty((0, 20), dtype=np.uint64))


def test_full_batch_cfg2_vs_c_oracle(eng):
    """BASELINE.json config 2: 2^16 variable-base MUL_endo, random 256-bit scalars (seeds of SURVEY 8d);
    every one of the 65536 raw R1 outputs compared with the C oracle, plus a Python-oracle sample."""
    n = 1 << 16
    s = seeded_scalars(20002, n)
    pts = torsion_points(eng, 20003, n)
    got = eng.mul_endo(s, pts)
    assert np.array_equal(got, oc.mul(oc.ENDO, s, pts))
    for i in range(0, n, n // 16):
        m = codec.unpack_scalars(s[i:i + 1])[0]
        assert codec.unpack_fp2s(got[i]) == o.MUL_endo(m, codec.unpack_fp2s(pts[i]))
    gotw = eng.mul_windowed(s[:8192], pts[:8192])
    assert np.array_equal(gotw, oc.mul(oc.WINDOWED, s[:8192], pts[:8192]))


def test_edge_scalars_through_the_one_lane_kernels(eng):
    """The reference's edge scalars reach the GPU suite on SMALL batches, i.e. on the two- and four-lane kernels.  The fused one-lane
    kernels -- round 6: digits as a nibble stream (recode_nibbles), 32-bit entry offsets, the signed store -- run from more than half a
    generation on: a whole generation whose head holds every edge pattern of the recoding (0, 1, 2, N-1, N, N+1, 2N, 2^255, 2^256-1,
    single words of ones, alternating bits, lone top and bottom bits of every 64-bit word), rest random; every output against the C
    oracle: MUL_endo, DH_endo (the same fused ladder, DH flavour), the mixed batch (mixed_queue_kernel's variable-base items) and the
    affine flavour (fused I/O flags)."""
    n = eng.lanes
    N = o.N
    edge = [0, 1, 2, N - 1, N, N + 1, 2 * N, 1 << 255, (1 << 256) - 1, (1 << 64) - 1, ((1 << 64) - 1) << 64, ((1 << 64) - 1) << 128, ((1 << 64) - 1) << 192,
            int("55" * 32, 16), int("aa" * 32, 16), (1 << 256) - 2, (1 << 255) - 1]
    edge += [1 << (64 * w) for w in range(4)] + [1 << (64 * w + 63) for w in range(4)] + [(1 << 256) - 1 - (1 << (64 * w)) for w in range(4)]
    s = seeded_scalars(4501, n)
    s[:len(edge)] = codec.pack_scalars(edge)
    pts = torsion_points(eng, 4502, n)
    want = oc.mul(oc.ENDO, s, pts)
    assert np.array_equal(eng.mul_endo(s, pts), want)
    aff = oc.r1_to_affine(pts)
    out, st = eng.dh_endo(s, aff)
    want_dh, want_st = oc.dh(oc.ENDO, s, aff)
    assert np.array_equal(st, want_st) and np.array_equal(out, want_dh)
    lifted = np.zeros((n, 20), dtype=np.uint64)
    lifted[:, 0:8] = aff; lifted[:, 8] = 1; lifted[:, 12:20] = aff
    assert np.array_equal(eng.mul_affine(s, aff), oc.r1_to_affine(oc.mul(oc.ENDO, s, lifted)))
    n2 = 2 * n                                                  # mixed: variable-base items at the head (flags 1), fixed-base behind
    s2 = np.concatenate([s, s]); p2 = np.concatenate([pts, pts])
    flags = np.concatenate([np.ones(n, dtype=np.uint8), np.zeros(n, dtype=np.uint8)])
    te = oc.table(oc.ENDO, codec.pack_point(G1))
    got = eng.mul_endo_mixed(s2, p2, flags, te)
    assert np.array_equal(got[:n], want) and np.array_equal(got[n:], oc.mul(oc.ENDO, s, None, te))


def test_full_batch_cfg3_fixed_base(eng):
    """BASELINE.json config 3: 2^20 fixed-base MUL_windowed(m, G, table): a 2^15 slice against the C oracle,
    and on the whole batch the size-independent property MUL_windowed == MUL_endo as affine points
    (two independent algorithms), checked on the GPU by cross-multiplication X1*Z2 == X2*Z1, Y1*Z2 == Y2*Z1."""
    n = 1 << 20
    s = seeded_scalars(30002, n)
    g = codec.pack_point(G1)
    tw, te = eng.table_windowed(g), eng.table_endo(g)
    w = eng.mul_windowed_fixed(s, tw)
    e = eng.mul_endo_fixed(s, te)
    k = 1 << 15
    assert np.array_equal(w[:k], oc.mul(oc.WINDOWED, s[:k], None, tw))
    assert np.array_equal(e[:k], oc.mul(oc.ENDO, s[:k], None, te))
    x1z2 = eng.prim("FP2_MUL", np.concatenate([w[:, 0:4], e[:, 8:12]], axis=1))
    x2z1 = eng.prim("FP2_MUL", np.concatenate([e[:, 0:4], w[:, 8:12]], axis=1))
    y1z2 = eng.prim("FP2_MUL", np.concatenate([w[:, 4:8], e[:, 8:12]], axis=1))
    y2z1 = eng.prim("FP2_MUL", np.concatenate([e[:, 4:8], w[:, 8:12]], axis=1))
    assert np.array_equal(x1z2, x2z1) and np.array_equal(y1z2, y2z1)


def test_linearity_property(eng):
    """[a]P + [b]P == [a+b]P (mod N) on 4096 random points: affine equality via the DH-free path."""
    n = 4096
    a, b = seeded_scalars(1, n), seeded_scalars(2, n)
    ai, bi = codec.unpack_scalars(a), codec.unpack_scalars(b)
    c = codec.pack_scalars([(x + y) % o.N for x, y in zip(ai, bi)])
    pts = torsion_points(eng, 3, n)
    pa, pb, pc = eng.mul_endo(a, pts), eng.mul_endo(b, pts), eng.mul_endo(c, pts)
    sm = eng.prim("PT_ADD", np.concatenate([pa, eng.prim("PT_R1TOR2", pb)], axis=1))
    assert np.array_equal(eng.prim("PT_R1TOAFFINE", sm), eng.prim("PT_R1TOAFFINE", pc))


def test_mixed_batch(eng):
    """BASELINE.json config 5 shape: element i fixed-base (flag 0) or variable-base (flag 1)."""
    n = 5000
    s = seeded_scalars(50001, n)
    pts = torsion_points(eng, 50003, n)
    flags = (np.frombuffer(random.Random(50002).getrandbits(8 * n).to_bytes(n, "little"), dtype=np.uint8) & 1).copy()
    tbl = eng.table_endo(codec.pack_point(G1))
    got = eng.mul_endo_mixed(s, pts, flags, tbl)
    want = np.where(flags[:, None] == 0, oc.mul(oc.ENDO, s, None, tbl), oc.mul(oc.ENDO, s, pts))
    assert np.array_equal(got, want)
    assert np.array_equal(eng.mul_endo_mixed(s, pts, np.zeros(n, np.uint8), tbl), oc.mul(oc.ENDO, s, None, tbl))
    assert np.array_equal(eng.mul_endo_mixed(s, pts, np.ones(n, np.uint8), tbl), oc.mul(oc.ENDO, s, pts))


def test_device_pointer_api_matches_host_api(eng):
    import torch
    n = 3000
    s, pts = seeded_scalars(9, n), torsion_points(eng, 10, n)
    ds = torch.from_numpy(s.view(np.int64)).cuda()
    dp = torch.from_numpy(pts.view(np.int64)).cuda()
    out = torch.empty((n, 20), dtype=torch.int64, device="cuda")
    eng.set_stream(torch.cuda.current_stream().cuda_stream)
    try:
        eng.mul_endo_dev(ds, dp, out, n)
        torch.cuda.synchronize()
    finally:
        eng.set_stream(None)
    assert np.array_equal(out.cpu().numpy().view(np.uint64), eng.mul_endo(s, pts))
    # device arrays must be 16-byte aligned (they are accessed as 16-byte vectors): a view 8 bytes in is refused
    from fourq_amd import FourQError
    flat = torch.empty(n * 20 + 2, dtype=torch.int64, device="cuda")
    with pytest.raises(FourQError):
        eng.mul_endo_dev(ds, dp, flat[1:], n)
    with pytest.raises(FourQError):
        eng.mul_endo_dev(ds.flatten()[1:], dp, out, n - 1)
    eng.mul_endo_dev(ds, dp, flat[2:], n)                      # 16 bytes in is fine
    torch.cuda.synchronize()
    assert np.array_equal(flat[2:].reshape(n, 20).cpu().numpy().view(np.uint64), out.cpu().numpy().view(np.uint64))


def test_full_size_cfg5_mixed_vs_c_oracle(eng):
    """BASELINE.json config 5 at its full size: 2^20 elements, 50% fixed-base / 50% variable-base by a
    seeded bitstream, every output compared with the C oracle."""
    n = 1 << 20                       # seeds as bench.py --workload cfg5 on rank 0 (flags 50002, points 50003; scalars 50004)
    s = seeded_scalars(50004, n)
    pts = torsion_points(eng, 50003, n)
    flags = (np.frombuffer(random.Random(50002).getrandbits(8 * n).to_bytes(n, "little"), dtype=np.uint8) & 1).copy()
    tbl = eng.table_endo(codec.pack_point(G1))
    got = eng.mul_endo_mixed(s, pts, flags, tbl)
    want = oc.mul(oc.ENDO, s, None, tbl)
    var = np.flatnonzero(flags)
    want[var] = oc.mul(oc.ENDO, s[var], pts[var])
    assert 0.49 < flags.mean() < 0.51 and np.array_equal(got, want)


@pytest.mark.parametrize("pair_max", ["one lane per element", "two at most", "four at most"])
def test_one_lane_and_two_lane_kernels_on_small_batches_and_tails(pair_max, monkeypatch):
    """Plain variable-base MUL_endo: batches of at most half a generation, and the tail of a batch past whole generations, run two
    lanes per element (pair.hip.h: real parts in even lanes, imaginary parts in odd lanes), those of at most a quarter generation FOUR
    (two pairs sharing the products of every formula level); FOURQ_PAIR_MAX=0 keeps everything on the one-lane fused kernel,
    FOURQ_QUAD_MAX=0 everything small on the two-lane kernels.  Same R1 tuples every way: edge scalars on G and -G, ragged sizes
    around the 64- and 128-element blocks, the switches at a quarter and at half a generation, and a tail behind one fused generation
    -- every output against the C oracle."""
    from fourq_amd import Engine
    monkeypatch.delenv("FOURQ_PAIR_MAX", raising=False)
    monkeypatch.delenv("FOURQ_QUAD_MAX", raising=False)
    if pair_max == "one lane per element":
        monkeypatch.setenv("FOURQ_PAIR_MAX", "0")
    elif pair_max == "two at most":
        monkeypatch.setenv("FOURQ_QUAD_MAX", "0")
    N = o.N
    edge = [0, 1, 2, N - 1, N, N + 1, 2 * N, 1 << 255, (1 << 256) - 1]
    negG = o.AffineToR1(o.GFp2.neg(o.Gx), o.Gy)
    with Engine(0) as e:
        lanes = e.lanes
        s = codec.pack_scalars(edge + edge)
        p = codec.pack_points([G1] * len(edge) + [negG] * len(edge), 5)
        assert codec.unpack_points(e.mul_endo(s, p)) == [o.MUL_endo(m, P) for m, P in zip(edge + edge, [G1] * len(edge) + [negG] * len(edge))]
        n = lanes + 300
        sc = seeded_scalars(61, n)
        pts = torsion_points(e, 62, n)
        want = oc.mul(oc.ENDO, sc, pts)
        for m in (1, 2, 63, 64, 65, 127, 128, 129, 255, 4097, lanes // 4, lanes // 4 + 1, lanes // 2, lanes // 2 + 1, lanes + 1, lanes + 300):
            assert np.array_equal(e.mul_endo(sc[:m], pts[:m]), want[:m]), (pair_max, m)
        e.ct_select = True                                        # the pair-lane kernels with the lane's table scanned at every step
        for m in (1, 129, 4097, lanes + 300):
            assert np.array_equal(e.mul_endo(sc[:m], pts[:m]), want[:m]), (pair_max, "ct", m)
        assert codec.unpack_points(e.mul_endo(s, p)) == [o.MUL_endo(m, P) for m, P in zip(edge + edge, [G1] * len(edge) + [negG] * len(edge))]
        e.ct_select = False
        # device flavour, a 16-byte-aligned offset into bigger arrays (the pair kernel's halves are 16-byte accesses)
        import torch
        ds, dp = torch.from_numpy(sc.view(np.int64)).cuda(), torch.from_numpy(pts.view(np.int64)).cuda()
        out = torch.zeros((n, 20), dtype=torch.int64, device="cuda")
        e.set_stream(torch.cuda.current_stream().cuda_stream)
        e.mul_endo_dev(ds[3:], dp[3:], out[3:], 1000)
        torch.cuda.synchronize()
        e.set_stream(None)
        got = out.cpu().numpy().view(np.uint64)
        assert np.array_equal(got[3:1003], want[3:1003]) and not got[:3].any() and not got[1003:].any()
        # the other variable-base entry points take the same route: MUL_windowed, DH_endo / DH_windowed with both rejections
        k = 5000
        want_w = oc.mul(oc.WINDOWED, sc[:k], pts[:k])
        aff = e.prim("PT_R1TOAFFINE", pts[:k]).copy()
        aff[7, 0] ^= 1                                            # not on the curve
        P392 = codec.pack_point(_kat_p392())
        aff[11] = P392                                            # order divides 392: the neutral point after cofactor clearing
        want_dh = {algo: oc.dh(algo, sc[:k], aff) for algo in (oc.ENDO, oc.WINDOWED)}
        for ct in (False, True):                                  # both selection modes of every variable-base kernel of the route
            e.ct_select = ct
            for m in (1, 129, k):
                assert np.array_equal(e.mul_windowed(sc[:m], pts[:m]), want_w[:m]), (pair_max, ct, m)
            for algo, fn in ((oc.ENDO, e.dh_endo), (oc.WINDOWED, e.dh_windowed)):
                want_d, want_st = want_dh[algo]
                for m in (1, 12, 129, k):
                    got_d, got_st = fn(sc[:m], aff[:m])
                    assert np.array_equal(got_st, want_st[:m]) and np.array_equal(got_d, want_d[:m]), (pair_max, ct, algo, m)
                assert want_st[7] == 1 and want_st[11] == 2
        e.ct_select = False
        # fixed base: the caller's table copied into the lanes' LDS rows (pair_kernel<..., FIXED>) or staged once per block (LDS ladders)
        te, tw = e.table_endo(codec.pack_point(G1)), e.table_windowed(codec.pack_point(G1))
        for ct in (False, True):
            e.ct_select = ct
            for m in (1, 129, k):
                assert np.array_equal(e.mul_endo_fixed(sc[:m], te), oc.mul(oc.ENDO, sc[:m], None, te)), (pair_max, ct, m)
                assert np.array_equal(e.mul_windowed_fixed(sc[:m], tw), oc.mul(oc.WINDOWED, sc[:m], None, tw)), (pair_max, ct, m)
            for algo, fn, tab in ((oc.ENDO, e.dh_endo, te), (oc.WINDOWED, e.dh_windowed, tw)):
                want_d, want_st = oc.dh(algo, sc[:k], aff, tab)
                for m in (12, k):
                    got_d, got_st = fn(sc[:m], aff[:m], tab)
                    assert np.array_equal(got_st, want_st[:m]) and np.array_equal(got_d, want_d[:m]), (pair_max, ct, algo, m)
                assert want_st[7] == 1
        e.ct_select = False


def _kat_p392():
    import json
    import os
    from conftest import GOLDEN, unhex
    with open(os.path.join(GOLDEN, "kat.json")) as fh:
        return unhex(json.load(fh)["P392"])


@pytest.mark.parametrize("queue", ["0", "1", None])
def test_mixed_batch_on_both_routes(queue, monkeypatch):
    """A mixed batch through round 2's three launches (FOURQ_MIXED_QUEUE=0), through the persistent work-queue kernel
    (FOURQ_MIXED_QUEUE=1: BASELINE config 5's mechanism) and by default routing -- at most half a generation: the two- / four-lane
    kernels with the table chosen per element (pair_kernel<..., MIXED>) -- both selection modes, ragged item counts of both kinds."""
    from fourq_amd import Engine
    if queue is None:
        monkeypatch.delenv("FOURQ_MIXED_QUEUE", raising=False)
    else:
        monkeypatch.setenv("FOURQ_MIXED_QUEUE", queue)
    n = 9001 if queue is not None else 40000                                       # default routing: four lanes, two lanes, the queue kernel
    s = seeded_scalars(50011, n)
    flags = (seeded_scalars(50012, n)[:, 0] % 3 == 0).astype(np.uint8)              # a third variable-base
    with Engine(0) as e:
        pts = torsion_points(e, 50013, n)
        tbl = e.table_endo(codec.pack_point(G1))
        want = np.where(flags[:, None] == 0, oc.mul(oc.ENDO, s, None, tbl), oc.mul(oc.ENDO, s, pts))
        for ct in (False, True):
            e.ct_select = ct
            assert np.array_equal(e.mul_endo_mixed(s, pts, flags, tbl), want), (queue, ct)
            for m in (1, 63, 64, 65) + ((127, 129, 9001, e.lanes // 4, e.lanes // 4 + 1, e.lanes // 2) if queue is None else ()):   # around the work item / the blocks / the route switches
                assert np.array_equal(e.mul_endo_mixed(s[:m], pts[:m], flags[:m], tbl), want[:m]), (queue, ct, m)
            if queue is None:                                                      # all of one kind, on the small-batch kernels
                m = 3000
                assert np.array_equal(e.mul_endo_mixed(s[:m], pts[:m], np.zeros(m, np.uint8), tbl), oc.mul(oc.ENDO, s[:m], None, tbl)), ct
                assert np.array_equal(e.mul_endo_mixed(s[:m], pts[:m], np.ones(m, np.uint8), tbl), oc.mul(oc.ENDO, s[:m], pts[:m])), ct


@pytest.mark.parametrize("extra", [0, 300, 9000])
def test_constant_time_mixed_round_cuts_a_small_remainder_off_the_fused_generations(extra):
    """Constant-time mode, a round larger than one generation: the variable-base ids past whole generations of the fused kernel
    ride with the fixed-base elements when there are at most lanes / 8 of them (mixed_ct_tail_kernel); 0 ids: nothing to cut,
    9 000: too many, a second fused generation.  Every output against the C oracle."""
    from fourq_amd import Engine
    with Engine(0) as e:
        e.ct_select = True
        n_var, n_fix = e.lanes + extra, 3000
        n = n_var + n_fix
        flags = np.ones(n, dtype=np.uint8)
        flags[np.random.RandomState(7).choice(n, n_fix, replace=False)] = 0         # the fixed-base elements scattered through the batch
        s = seeded_scalars(50021 + extra, n)
        pts = torsion_points(e, 50023, n)
        tbl = e.table_endo(codec.pack_point(G1))
        want = oc.mul(oc.ENDO, s, pts)
        fix = np.flatnonzero(flags == 0)
        want[fix] = oc.mul(oc.ENDO, s[fix], None, tbl)
        assert np.array_equal(e.mul_endo_mixed(s, pts, flags, tbl), want)


@pytest.mark.parametrize("n", [131071, 131072, 262144 + 77])
def test_prep_plus_ladder_route_boundaries(n, monkeypatch):
    """Large variable-base MUL_windowed / DH batches take the two-kernel route (prep_kernel +
    ladder_kernel<PREBUILT>) in chunks of the resident lane count; FOURQ_SPLIT_ALL=1 sends MUL_endo through it
    too.  Sizes just below / at the switch and just past one chunk; outputs vs the C oracle."""
    from fourq_amd import Engine
    monkeypatch.setenv("FOURQ_SPLIT_ALL", "1")
    with Engine(0) as eng:
        s = seeded_scalars(4100 + n % 97, n)
        pts = torsion_points(eng, 4200 + n % 89, n)
        assert np.array_equal(eng.mul_endo(s, pts), oc.mul(oc.ENDO, s, pts))
        k = 20000
        assert np.array_equal(eng.mul_windowed(s, pts)[-k:], oc.mul(oc.WINDOWED, s[-k:], pts[-k:]))


def test_special_base_points(eng, golden):
    """Bases the formulas must still follow the reference on: the neutral point, -G, a point of order dividing 392
    (MUL_* give 'wrong' answers off the N-torsion, curve4q.py docstring of the draft; parity is with the reference's)."""
    from conftest import unhex
    P392 = unhex(golden("kat.json", raw=True)["P392"])
    bases = [o.AffineToR1(o.Ox, o.Oy), o.AffineToR1(o.f2_neg(o.Gx), o.Gy), o.AffineToR1(*P392),
             (o.Gx, o.Gy, (1, 0), (0, 0), (0, 0)), ((0, 0), (0, 0), (0, 0), (0, 0), (0, 0))]
    rng = random.Random(321)
    ms = [0, 1, 2, 391, 392, o.N, (1 << 256) - 1] + [rng.getrandbits(256) for _ in range(5)]
    rows = [(m, P) for P in bases for m in ms]
    s = codec.pack_scalars([m for m, _ in rows])
    p = codec.pack_points([P for _, P in rows], 5)
    assert codec.unpack_points(eng.mul_endo(s, p)) == [o.MUL_endo(m, P) for m, P in rows]
    assert codec.unpack_points(eng.mul_windowed(s, p)) == [o.MUL_windowed(m, P) for m, P in rows]


def test_two_engines_and_reuse_after_error():
    """Contexts are independent; an invalid call leaves a context usable."""
    from fourq_amd import Engine, FourQError
    s, g = seeded_scalars(5, 300), codec.pack_point(G1)
    with Engine(0) as e1, Engine(0) as e2:
        t1, t2 = e1.table_endo(g), e2.table_windowed(g)
        a = e1.mul_endo_fixed(s, t1)
        b = e2.mul_windowed_fixed(s, t2)
        with pytest.raises(ValueError):
            e1.mul_endo(s, np.zeros((7, 20), dtype=np.uint64))
        with pytest.raises(FourQError):
            e1.prim(999, np.zeros((1, 4), dtype=np.uint64))
        assert np.array_equal(e1.mul_endo_fixed(s, t1), a) and np.array_equal(e2.mul_windowed_fixed(s, t2), b)
        assert np.array_equal(a, oc.mul(oc.ENDO, s, None, t1)) and np.array_equal(b, oc.mul(oc.WINDOWED, s, None, t2))


def test_ct_select_from_the_environment(monkeypatch):
    """FOURQ_CT_SELECT=1 at context creation turns the constant-time selection on for every entry point of that context."""
    from fourq_amd import Engine
    monkeypatch.setenv("FOURQ_CT_SELECT", "1")
    with Engine(0) as e:
        assert e.ct_select
        s = seeded_scalars(77, 700)
        pts = torsion_points(e, 78, 700)
        assert np.array_equal(e.mul_endo(s, pts), oc.mul(oc.ENDO, s, pts))
        e.ct_select = False
        assert not e.ct_select and np.array_equal(e.mul_windowed(s, pts), oc.mul(oc.WINDOWED, s, pts))
    monkeypatch.delenv("FOURQ_CT_SELECT")
    with Engine(0) as e:
        assert not e.ct_select


def test_batches_beyond_the_abi_limit_are_refused(eng):
    """n > FOURQ_MAX_BATCH (the kernels' 32-bit counters round n up to whole 256-lane blocks) is FOURQ_ERR_INVALID before
    anything is read or launched; the context stays usable."""
    import ctypes
    from fourq_amd import _lib
    lib = _lib.load()
    s, pts = seeded_scalars(3, 16), torsion_points(eng, 4, 16)
    out = np.empty((16, 20), dtype=np.uint64)
    ptr = lambda a: ctypes.c_void_p(a.ctypes.data)
    for n in (_lib.MAX_BATCH + 1, (1 << 32) - 1, 1 << 40):
        assert lib.fourq_mul_endo_batch(eng._ctx, ptr(s), ptr(pts), ptr(out), n) == _lib.ERR_INVALID
        assert lib.fourq_mul_endo_batch_dev(eng._ctx, ptr(s), ptr(pts), ptr(out), n) == _lib.ERR_INVALID
        st = np.empty(16, dtype=np.uint8)
        assert lib.fourq_dh_endo_batch(eng._ctx, ptr(s), ptr(pts), None, ptr(out), ptr(st), n) == _lib.ERR_INVALID
    assert np.array_equal(eng.mul_endo(s, pts), oc.mul(oc.ENDO, s, pts))


def test_dev_entry_points_can_be_captured_into_a_hip_graph(eng):
    """The _dev entry points only enqueue kernels on the context's stream, so a caller can capture them into a HIP graph
    (torch.cuda.CUDAGraph) and replay it: here a variable-base MUL_endo followed by a fixed-base one that overwrites the first
    one's input, replayed twice.  (Call each route once before capturing: the first call stages tables / sizes buffers.)"""
    import torch
    dev = torch.device("cuda", 0)
    n = 3000
    s_h, k_h = seeded_scalars(61, n), seeded_scalars(62, n)
    te = oc.table(oc.ENDO, codec.pack_point(G1))
    p_h = oc.mul(oc.ENDO, k_h, None, te)
    s = torch.from_numpy(s_h.view(np.int64)).to(dev)
    p = torch.from_numpy(p_h.view(np.int64)).to(dev)
    out = torch.empty((n, 20), dtype=torch.int64, device=dev)
    side = torch.cuda.Stream(device=dev)
    eng.set_stream(side.cuda_stream)
    try:
        eng.mul_endo_fixed_dev(s, te, out, n)              # stages the table outside the capture
        eng.sync()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.stream(side):
            graph.capture_begin()
            eng.mul_endo_dev(s, p, out, n)
            eng.mul_endo_fixed_dev(s, te, p, n)
            graph.capture_end()
        out.zero_()
        graph.replay()
        torch.cuda.synchronize()
        assert np.array_equal(out.cpu().numpy().view(np.uint64), oc.mul(oc.ENDO, s_h, p_h))
        p2 = p.cpu().numpy().view(np.uint64).copy()        # the replay left [s_i]G in `p`
        assert np.array_equal(p2, oc.mul(oc.ENDO, s_h, None, te))
        graph.replay()
        torch.cuda.synchronize()
        assert np.array_equal(out.cpu().numpy().view(np.uint64), oc.mul(oc.ENDO, s_h, p2))
    finally:
        eng.set_stream(None)
