#!/usr/bin/env python3
"""Differential fuzz of the batched entry points against the C oracle: random batch sizes around every route
boundary, random routing knobs, rejected points sprinkled in.  Test infrastructure (uses oracle/); GPU box.
    python tools/fuzz.py [seconds]"""
import os
os.environ.setdefault("FOURQ_DEBUG_ROUTES", "1")      # the FOURQ_* route hooks below are read only under this gate (tools/README.md)
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import curve4q_oracle as o
import oracle_c as oc
from fourq_amd import Engine, codec

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = random.Random(int(os.environ.get("FUZZ_SEED", "12345")))
G1 = o.AffineToR1(o.Gx, o.Gy)
g1 = codec.pack_point(G1)
gaff = codec.pack_point((o.Gx, o.Gy))
p392 = None


def scalars(n):
    return np.frombuffer(rng.getrandbits(256 * n).to_bytes(32 * n, "little"), dtype="<u8").reshape(n, 4).copy()


def pick_n(lanes):
    edges = [1, 2, 63, 64, 65, 127, 128, 129, 255, 256, 257, 4095, 4096, 4097, lanes // 4 - 1, lanes // 4, lanes // 4 + 1, lanes // 2, lanes // 2 + 1,
             lanes - 1, lanes, lanes + 1, lanes + 64, lanes + lanes // 4, lanes + lanes // 4 + 1, 2 * lanes - 1, 2 * lanes, 2 * lanes + 1, 4 * lanes + 3,
             5 * lanes, 7 * lanes + 5]                      # round 5: enough generations for the host-array pipeline to reuse its slots
    return rng.choice(edges) if rng.random() < 0.5 else rng.randrange(1, 3 * lanes)


t_end = time.time() + budget
rounds = 0
while time.time() < t_end:
    knobs = {}
    if rng.random() < 0.7:
        knobs["FOURQ_SPLIT_MIN"] = str(rng.choice([1, 256, 1000, 70000]))
        knobs["FOURQ_SPLIT_CHUNK"] = str(rng.choice([256, 4096, 65536, 262144]))
        knobs["FOURQ_NORM_K"] = str(rng.choice([0, 2, 4, 8]))
        if rng.random() < 0.3:
            knobs["FOURQ_SPLIT_ALL"] = "1"
        if rng.random() < 0.3:
            knobs["FOURQ_SPLIT_ENDO_MIN"] = str(rng.choice([0, 300, 70000]))
    if rng.random() < 0.35:
        knobs["FOURQ_CT_SELECT"] = "1"
    if rng.random() < 0.3:
        knobs["FOURQ_HOST_BOUNCE"] = "0"
    if rng.random() < 0.4:
        knobs["FOURQ_MIXED_QUEUE"] = rng.choice(["0", "1"])          # round 3: the persistent work-queue kernel forced on / off
    if rng.random() < 0.3:
        knobs["FOURQ_PAIR_MAX"] = rng.choice(["0", "100", "5000"])     # round 3: the two-lanes-per-element kernel off / for tiny tails only
    if rng.random() < 0.3:
        knobs["FOURQ_HOST_ZERO_COPY"] = "0"                              # tiny host calls through hipMemcpyAsync instead of pinned memory read and written in place
    if rng.random() < 0.3:
        knobs["FOURQ_QUAD_MAX"] = rng.choice(["0", "64", "1000"])      # the four-lanes-per-element kernels off / for tiny batches and tails only
    if rng.random() < 0.5:                                             # round 5: the shape of the host-array pipeline
        knobs["FOURQ_PIPE_SLOTS"] = rng.choice(["2", "3", "4", "6"])
        knobs["FOURQ_PIPE_GENS"] = rng.choice(["0", "1", "2", "3"])
        if rng.random() < 0.3:
            knobs["FOURQ_PIPE_HOST_WAIT"] = "1"
    if rng.random() < 0.25:                                            # round 6: the affine / encoded flavours through lift + R1 rows instead of the fused I/O flags
        knobs["FOURQ_FUSED_IO"] = "0"
    if rng.random() < 0.25:                                            # ... and chunk plans from the compiled-in guesses instead of the context's measurements
        knobs["FOURQ_PIPE_MEASURE"] = "0"
    for k in ("FOURQ_FUSED_IO", "FOURQ_PIPE_MEASURE", "FOURQ_PIPE_SLOTS", "FOURQ_PIPE_GENS", "FOURQ_PIPE_HOST_WAIT", "FOURQ_PAIR_MAX", "FOURQ_QUAD_MAX", "FOURQ_HOST_ZERO_COPY", "FOURQ_SPLIT_MIN", "FOURQ_SPLIT_CHUNK", "FOURQ_NORM_K", "FOURQ_SPLIT_ALL", "FOURQ_SPLIT_ENDO_MIN", "FOURQ_CT_SELECT", "FOURQ_HOST_BOUNCE", "FOURQ_MIXED_QUEUE"):
        os.environ.pop(k, None)
    os.environ.update(knobs)
    with Engine(0) as eng:
        te, tw = eng.table_endo(g1), eng.table_windowed(g1)
        if p392 is None:                                   # a point of small order (neutral after cofactor clearing)
            import json
            kat = json.load(open(os.path.join(ROOT, "tests", "golden", "kat.json")))

            def unhex(v):
                return tuple(unhex(x) for x in v) if isinstance(v, (list, tuple)) else int(v, 16)
            p392 = codec.pack_point(unhex(kat["P392"]))
        for _ in range(6):
            n = pick_n(eng.lanes)
            s = scalars(n)
            if rng.random() < 0.2:
                s[rng.randrange(n)] = 0
            pts = eng.mul_endo_fixed(scalars(n), te)
            what = rng.choice(["endo", "win", "endo_fixed", "win_fixed", "mixed", "dh_endo", "dh_win", "dh_fixed", "comb", "pinned", "dh_bytes", "exchange", "mul_affine", "mul_bytes"])
            if what == "endo":
                ok = np.array_equal(eng.mul_endo(s, pts), oc.mul(oc.ENDO, s, pts))
            elif what == "pinned":                           # the same from pinned host arrays (DMA in place, no bounce)
                sp, pp, op_ = eng.host_array(s), eng.host_array(pts), eng.host_empty((n, 20))
                which = rng.choice(["win", "endo", "win_fixed"])
                if which == "win":
                    ok = np.array_equal(eng.mul_windowed(sp, pp, out=op_), oc.mul(oc.WINDOWED, s, pts))
                elif which == "endo":
                    ok = np.array_equal(eng.mul_endo(sp, pp, out=op_), oc.mul(oc.ENDO, s, pts))
                else:
                    ok = np.array_equal(eng.mul_windowed_fixed(sp, tw, out=op_), oc.mul(oc.WINDOWED, s, None, tw))
                for a_ in (sp, pp, op_):
                    eng.host_free(a_)
            elif what in ("dh_bytes", "exchange"):
                gg = np.repeat(gaff.reshape(1, 8), n, axis=0)
                pub, st0 = oc.dh(oc.ENDO, scalars(n), gg)   # public keys DH(b, G)
                if what == "dh_bytes":
                    keys = eng.encode(pub)
                    if n > 3:
                        keys[rng.randrange(n), 15] |= 0x80  # a reserved bit: decode status 1
                    got, gst = eng.dh_bytes(s, keys)
                    dec, dst = eng.decode(keys)
                    want, wst = oc.dh(oc.ENDO, s, dec)
                    wst = np.where(dst != 0, 16 + dst, wst).astype(np.uint8)
                    wenc = eng.encode(want)
                    wenc[wst != 0] = 0
                    ok = np.array_equal(got, wenc) and np.array_equal(gst, wst)
                else:
                    b2 = scalars(n)
                    mid, s1 = oc.dh(oc.ENDO, b2, gg)
                    want, s2 = oc.dh(oc.ENDO, s, mid)
                    wst = np.where(s1 != 0, s1, s2)
                    want[wst != 0] = 0
                    got, gst = eng.dh_exchange(s, b2, gaff)
                    ok = np.array_equal(got, want) and np.array_equal(gst, wst)
            elif what in ("mul_affine", "mul_bytes"):        # round 4: MUL_* with affine / encoded I/O, both algorithms
                kind, okind = rng.choice([("endo", oc.ENDO), ("windowed", oc.WINDOWED)])
                aff_in = oc.r1_to_affine(pts)
                lifted = np.zeros((n, 20), dtype=np.uint64)
                lifted[:, 0:8] = aff_in; lifted[:, 8] = 1; lifted[:, 12:20] = aff_in
                want = oc.r1_to_affine(oc.mul(okind, s, lifted))
                if what == "mul_affine":
                    ok = np.array_equal(eng.mul_affine(s, aff_in, kind=kind), want)
                else:
                    keys = oc.encode(aff_in).copy()
                    wenc, wst = oc.encode(want).copy(), np.zeros(n, dtype=np.uint8)
                    if n > 3:
                        j = rng.randrange(n)
                        keys[j, 15] |= 0x80                  # a reserved bit: decode status 1
                        wenc[j], wst[j] = 0, 16 + 1
                    got, gst = eng.mul_bytes(s, keys, kind=kind)
                    ok = np.array_equal(got, wenc) and np.array_equal(gst, wst)
            elif what == "win":
                ok = np.array_equal(eng.mul_windowed(s, pts), oc.mul(oc.WINDOWED, s, pts))
            elif what == "endo_fixed":
                ok = np.array_equal(eng.mul_endo_fixed(s, te), oc.mul(oc.ENDO, s, None, te))
            elif what == "win_fixed":
                ok = np.array_equal(eng.mul_windowed_fixed(s, tw), oc.mul(oc.WINDOWED, s, None, tw))
            elif what == "mixed":
                flags = np.frombuffer(rng.getrandbits(8 * n).to_bytes(n, "little"), dtype=np.uint8) & rng.choice([1, 1, 0, 255])
                want = np.where(flags.reshape(-1, 1) != 0, oc.mul(oc.ENDO, s, pts), oc.mul(oc.ENDO, s, None, te))
                ok = np.array_equal(eng.mul_endo_mixed(s, pts, flags.copy(), te), want)
            else:
                aff = eng.prim("PT_R1TOAFFINE", pts)
                for _bad in range(rng.randrange(0, 4)):
                    j = rng.randrange(n)
                    if rng.random() < 0.5:
                        aff[j, 0] ^= np.uint64(1)          # off the curve
                    else:
                        aff[j] = p392                      # small order -> neutral
                if what == "dh_endo":
                    got, want = eng.dh_endo(s, aff), oc.dh(oc.ENDO, s, aff)
                elif what == "dh_win":
                    got, want = eng.dh_windowed(s, aff), oc.dh(oc.WINDOWED, s, aff)
                elif what == "dh_fixed":
                    t392 = eng.table_endo(codec.pack_point(o.MUL_endo(392, G1)))
                    gg = np.repeat(gaff.reshape(1, 8), n, axis=0)
                    got, want = eng.dh_endo(s, gg, t392), oc.dh(oc.ENDO, s, gg)
                else:
                    comb = eng.comb_table(codec.pack_point(o.MUL_endo(392, G1)))
                    gg = np.repeat(gaff.reshape(1, 8), n, axis=0)
                    got, want = eng.comb_mul(s, comb), oc.dh(oc.ENDO, s, gg)
                ok = np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])
            rounds += 1
            print("%-10s n=%-7d knobs=%s %s" % (what, n, knobs, "ok" if ok else "MISMATCH"), flush=True)
            if not ok:
                sys.exit(1)
            if time.time() > t_end:
                break
print("FUZZ OK: %d rounds" % rounds)
