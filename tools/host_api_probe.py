#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-pointer ABI (H2D + kernel + D2H), for DESIGN.md section 6."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bench import seeded_scalars
from fourq_amd import Engine, codec, constants
eng = Engine(0)
g1 = codec.pack_point((constants.Gx, constants.Gy, (1, 0), constants.Gx, constants.Gy))
te = eng.table_endo(g1)
for lg in (16, 20):
    n = 1 << lg
    s = seeded_scalars(1, n)
    pts = eng.mul_endo_fixed(seeded_scalars(2, n), te)
    eng.mul_endo(s, pts)
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter(); eng.mul_endo(s, pts); best = min(best, time.perf_counter() - t0)
    print("host-pointer MUL_endo n=2^%d: %.3f ms -> %.1f Mmults/s (PCIe-inclusive, pageable host memory)" % (lg, best * 1e3, n / best / 1e6))
