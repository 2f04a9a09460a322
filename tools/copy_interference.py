#!/usr/bin/env python3
"""Does a host copy in flight slow the headline kernel down?  The fused MUL_endo kernel (2^16 elements, one generation) on one stream, timed by
events, alone and with a stream of 10 MiB copies (device->host, host->device, both) running beside it on other streams -- the situation of
every chunk of the host-array pipeline but the first and the last (tools/pipeline_probe.py)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from bench import seeded_scalars
from fourq_amd import Engine, codec, constants

dev = torch.device("cuda", 0)
sa, sb, sc = torch.cuda.Stream(dev), torch.cuda.Stream(dev), torch.cuda.Stream(dev)
n = 1 << 16
g1 = codec.pack_point((constants.Gx, constants.Gy, (1, 0), constants.Gx, constants.Gy))
with Engine(0, stream=sa.cuda_stream) as eng:
    te = eng.table_endo(g1)
    with torch.cuda.stream(sa):
        s = torch.from_numpy(seeded_scalars(1, n).view(np.int64)).to(dev)
        k = torch.from_numpy(seeded_scalars(2, n).view(np.int64)).to(dev)
        pts = torch.empty((n, 20), dtype=torch.int64, device=dev)
        out = torch.empty((n, 20), dtype=torch.int64, device=dev)
        eng.mul_endo_fixed_dev(k, te, pts, n)
    torch.cuda.synchronize()
    MB = 10 << 20
    h_out = torch.empty(MB, dtype=torch.uint8).pin_memory(); d_src = torch.empty(MB, dtype=torch.uint8, device=dev)
    h_in = torch.empty(MB, dtype=torch.uint8).pin_memory(); d_dst = torch.empty(MB, dtype=torch.uint8, device=dev)

    def run(d2h, h2d, launches=300):
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2 * launches)]
        torch.cuda.synchronize()
        for i in range(launches):
            if d2h:
                with torch.cuda.stream(sb): h_out.copy_(d_src, non_blocking=True)
            if h2d:
                with torch.cuda.stream(sc): d_dst.copy_(h_in, non_blocking=True)
            ev[2 * i].record(sa)
            eng.mul_endo_dev(s, pts, out, n)
            ev[2 * i + 1].record(sa)
        torch.cuda.synchronize()
        t = sorted(ev[2 * i].elapsed_time(ev[2 * i + 1]) for i in range(100, launches))
        return t[len(t) // 2], t[0], t[-1]

    run(0, 0)                                   # clocks up
    for name, a, b in (("alone", 0, 0), ("beside device->host copies", 1, 0), ("beside host->device copies", 0, 1), ("beside both", 1, 1), ("alone again", 0, 0)):
        med, lo, hi = run(a, b)
        print("kernel %-28s median %.4f ms  (min %.4f, max %.4f)" % (name, med, lo, hi), flush=True)
