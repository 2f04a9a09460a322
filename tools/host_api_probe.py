#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-array ABI (H2D + kernels + D2H, pipelined over chunks), for DESIGN.md section 6:
pinned caller arrays, pageable arrays through the bounce slots, pageable arrays handed to hipMemcpyAsync directly
(FOURQ_HOST_BOUNCE=0).  Outputs are preallocated so that first-touch page faults are not part of the figure."""
import os, sys, time
os.environ.setdefault("FOURQ_DEBUG_ROUTES", "1")      # the FOURQ_* route hooks below are read only under this gate (tools/README.md)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bench import seeded_scalars
from fourq_amd import Engine, codec, constants

g1 = codec.pack_point((constants.Gx, constants.Gy, (1, 0), constants.Gx, constants.Gy))
for mode in ("pinned", "bounce", "direct"):
    os.environ["FOURQ_HOST_BOUNCE"] = "0" if mode == "direct" else "1"
    with Engine(0) as eng:
        te = eng.table_endo(g1)
        tw = eng.table_windowed(g1)
        for lg in (16, 18, 20):
            n = 1 << lg
            put = eng.host_array if mode == "pinned" else (lambda a: a)
            s = put(seeded_scalars(1, n))
            pts = put(eng.mul_endo_fixed(seeded_scalars(2, n), te))
            out = eng.host_empty((n, 20)) if mode == "pinned" else np.zeros((n, 20), dtype=np.uint64)
            for name, fn in (("MUL_endo var", lambda: eng.mul_endo(s, pts, out=out)), ("MUL_windowed fixed", lambda: eng.mul_windowed_fixed(s, tw, out=out))):
                fn()
                best = 1e9
                for _ in range(5):
                    t0 = time.perf_counter(); fn(); best = min(best, time.perf_counter() - t0)
                st = eng.host_stats()
                print("%-8s %-18s n=2^%d: %8.3f ms -> %6.1f Mmults/s  (chunks %d, copies %.1f / %.1f GB/s)" % (
                    mode, name, lg, best * 1e3, n / best / 1e6, st["chunks"], st["gbs_h2d"] or 0, st["gbs_d2h"] or 0), flush=True)
