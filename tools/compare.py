#!/usr/bin/env python3
"""Op-count and timing tables in the shape of the reference's impl/compare.py (:51-169, :171-219), for
the Curve4Q half only.

* op counts: GF(p^2) multiplications / squarings / additions / inversions per primitive, counted by
  wrapping the ORACLE's field functions (test infrastructure: this tool lives outside the product);
* timings: the pure-Python oracle on one core next to the MI355X engine (batched), per operation.

    python tools/compare.py            # op counts (CPU only)
    python tools/compare.py --gpu      # + timing table (needs an MI355X)
"""
import argparse
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import curve4q_oracle as o  # noqa: E402


class Counter:
    """Counts the oracle's GF(p^2) operations the way fields.py:156-199 does (conj = half an addition,
    an inversion's internal product is not a multiplication)."""

    NAMES = {"f2_mul": "M", "f2_sqr": "S", "f2_add": "A", "f2_sub": "A", "f2_neg": "A", "f2_conj": "C", "f2_inv": "I"}

    def __init__(self):
        self.n = {"M": 0, "S": 0, "A": 0.0, "I": 0}
        self._saved = {}

    def __enter__(self):
        for name, key in self.NAMES.items():
            fn = getattr(o, name)
            self._saved[name] = fn

            def wrapped(*a, _fn=fn, _key=key):
                if _key == "C":
                    self.n["A"] += 0.5
                elif _key == "I":
                    self.n["I"] += 1
                    self.n["M"] -= 1      # fields.py:197: the final product is part of the inversion
                    self.n["A"] -= 0.5
                else:
                    self.n[_key] += 1
                return _fn(*a)
            setattr(o, name, wrapped)
        return self

    def __exit__(self, *exc):
        for name, fn in self._saved.items():
            setattr(o, name, fn)


def count(fn, *args):
    with Counter() as c:
        fn(*args)
    return c.n["M"], c.n["S"], c.n["A"], c.n["I"]


def op_table(seed=1):
    rng = random.Random(seed)
    G = o.AffineToR1(o.Gx, o.Gy)
    m = rng.getrandbits(256) | 1
    P = o.MUL_endo(rng.getrandbits(256), G)
    T2 = o.R1toR2(P)
    t = o.tau(P[:3])
    rows = [
        ("DBL", count(o.DBL, P)), ("ADD", count(o.ADD, P, T2)), ("ADD_core", count(o.ADD_core, o.R1toR3(P), T2)),
        ("R1toR2", count(o.R1toR2, P)), ("R1toR3", count(o.R1toR3, P)), ("R2toR4", count(o.R2toR4, T2)),
        ("tau", count(o.tau, P[:3])), ("tau_dual", count(o.tau_dual, t)), ("upsilon", count(o.upsilon, t)), ("chi", count(o.chi, t)),
        ("phi", count(o.phi, P)), ("psi", count(o.psi, P)),
        ("table_windowed", count(o.table_windowed, P)), ("table_endo", count(o.table_endo, P)),
        ("MUL_windowed", count(o.MUL_windowed, m, P)), ("MUL_windowed(table)", count(o.MUL_windowed, m, P, o.table_windowed(P))),
        ("MUL_endo", count(o.MUL_endo, m, P)), ("MUL_endo(table)", count(o.MUL_endo, m, P, o.table_endo(P))),
        ("R1toAffine", count(o.R1toAffine, P)),
        ("DH_windowed", count(o.DH_windowed, m, (o.Gx, o.Gy))), ("DH_endo", count(o.DH_endo, m, (o.Gx, o.Gy))),
    ]
    return rows


def print_ops(rows):
    print("%-22s %8s %8s %8s %4s" % ("GF(p^2) ops", "M", "S", "A", "I"))
    for name, (M, S, A, I) in rows:
        print("%-22s %8d %8d %8g %4d" % (name, M, S, A, I))


def timing_table(n=1 << 16, cpu_ops=20, out=print):
    """The reference's timing table (compare.py:171-219) with the MI355X engine in the second column.  Every GPU result
    whose operation is timed is also CHECKED: the first `cpu_ops` elements of each batch are the operations the oracle
    is timed on, and their outputs must be bit-identical.  Returns [(name, oracle s/op, GPU s/op, elements checked)]."""
    import numpy as np
    from fourq_amd import Engine, codec
    G1, Gaff = o.AffineToR1(o.Gx, o.Gy), (o.Gx, o.Gy)
    raw = np.frombuffer(random.Random(3).getrandbits(256 * n).to_bytes(32 * n, "little"), dtype="<u8").reshape(n, 4).copy()
    ms = codec.unpack_scalars(raw[:cpu_ops])
    rows = []
    with Engine(0) as eng:
        g1 = codec.pack_point(G1)
        te, tw = eng.table_endo(g1), eng.table_windowed(g1)
        pts = eng.mul_endo_fixed(random_scalars(4, n), te)
        Ps = codec.unpack_points(pts[:cpu_ops])
        gaff = np.repeat(codec.pack_point(Gaff).reshape(1, 8), n, axis=0)
        Tw, Te = o.table_windowed(G1), o.table_endo(G1)
        dh = lambda fn: (lambda: fn(raw, gaff)[0])
        cases = [
            ("MUL_windowed(m,P)", lambda m, P: o.MUL_windowed(m, P), lambda: eng.mul_windowed(raw, pts)),
            ("MUL_windowed(m,G,table)", lambda m, P: o.MUL_windowed(m, G1, Tw), lambda: eng.mul_windowed_fixed(raw, tw)),
            ("MUL_endo(m,P)", lambda m, P: o.MUL_endo(m, P), lambda: eng.mul_endo(raw, pts)),
            ("MUL_endo(m,G,table)", lambda m, P: o.MUL_endo(m, G1, Te), lambda: eng.mul_endo_fixed(raw, te)),
            ("DH_windowed(m,G)", lambda m, P: o.DH_windowed(m, Gaff), dh(eng.dh_windowed)),
            ("DH_endo(m,G)", lambda m, P: o.DH_endo(m, Gaff), dh(eng.dh_endo)),
        ]
        out("%-26s %16s %22s %10s" % ("operation", "oracle ms/op (1 core)", "MI355X us/op (batch 2^%d, PCIe incl.)" % (n.bit_length() - 1), "ratio"))
        for name, cpu, gpu in cases:
            want, per_op = [], []
            for m, P in zip(ms, Ps):                 # median of the per-operation times: one descheduled call (a fresh
                t0 = time.perf_counter()             # box still paging its image in) must not decide the row
                want.append(cpu(m, P))
                per_op.append(time.perf_counter() - t0)
            c = sorted(per_op)[len(per_op) // 2]
            gpu()
            t0 = time.perf_counter()
            got = gpu()
            g = (time.perf_counter() - t0) / n
            if codec.unpack_points(got[:cpu_ops]) != want:
                raise SystemExit("PARITY FAILURE in the timing table: %s" % name)
            out("%-26s %16.3f %22.4f %10.0f" % (name, c * 1e3, g * 1e6, c / g))
            rows.append((name, c, g, len(want)))
    return rows


def random_scalars(seed, n):
    import numpy as np
    return np.frombuffer(random.Random(seed).getrandbits(256 * n).to_bytes(32 * n, "little"), dtype="<u8").reshape(n, 4).copy()


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpu", action="store_true")
    args = ap.parse_args()
    print_ops(op_table())
    if args.gpu:
        print()
        timing_table()
