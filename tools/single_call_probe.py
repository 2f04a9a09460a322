#!/usr/bin/env python3
"""Wall-clock cost of ONE reference-shaped call (a batch of one) through fourq_amd.curve4q and through the array API (GPU box)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from fourq_amd import curve4q, Engine, codec, default_engine

G = curve4q.AffineToR1(curve4q.Gx, curve4q.Gy)
m = 0x1234567890abcdef1234567890abcdef1234567890abcdef1234567890abcdef
P = curve4q.MUL_endo(m, G)
Q = curve4q.R1toAffine(curve4q.MUL_endo(7, G))          # a public key that is not the generator (DH on the generator itself is key generation: the comb)
for name, fn in (("curve4q.MUL_endo(m, P)", lambda: curve4q.MUL_endo(m, P)), ("curve4q.MUL_windowed(m, P)", lambda: curve4q.MUL_windowed(m, P)),
                 ("curve4q.DH_endo(m, Q)", lambda: curve4q.DH_endo(m, Q)),
                 ("curve4q.DH_endo(m, G) [comb]", lambda: curve4q.DH_endo(m, (curve4q.Gx, curve4q.Gy)))):
    for _ in range(20): fn()
    t0 = time.perf_counter()
    for _ in range(300): fn()
    print("%-28s %.3f ms per call" % (name, (time.perf_counter() - t0) / 300 * 1e3))
eng = default_engine()
s = codec.pack_scalars([m]); p = codec.pack_points([P], 5)
for n in (1, 64, 1024, 16384):
    ss, pp = np.repeat(s, n, 0), np.repeat(p, n, 0)
    for _ in range(10): eng.mul_endo(ss, pp)
    t0 = time.perf_counter()
    for _ in range(100): eng.mul_endo(ss, pp)
    print("Engine.mul_endo n=%-6d      %.3f ms per call" % (n, (time.perf_counter() - t0) / 100 * 1e3))

# the same single calls from 16 threads: combined into batches (fourq_amd/combine.py) vs one launch per call taking turns
import threading
for combine in (False, True):
    curve4q._COMBINE = combine
    T, R = 16, 200
    def work():
        for _ in range(R): curve4q.MUL_endo(m, P)
    work()
    ths = [threading.Thread(target=work) for _ in range(T)]
    t0 = time.perf_counter()
    for th in ths: th.start()
    for th in ths: th.join()
    dt = time.perf_counter() - t0
    print("curve4q.MUL_endo from %d threads, combine %-5s  %8.0f calls/s  (%.3f ms per call seen by a thread)" % (T, combine, T * R / dt, dt / R * 1e3))
print("combine_stats", curve4q.combine_stats())
