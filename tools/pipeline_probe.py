#!/usr/bin/env python3
"""One host-array call of each I/O format at 2^20 elements on pinned arrays, for a rocprofv3 --kernel-trace --memory-copy-trace timeline
(tools/runs/r04_pipeline_trace.sh): where the chunks' copies and kernels sit relative to each other.  Also the link itself: one direction
alone and both directions at once (torch copies on two streams), to tell a copy-bound pipeline from a bubble-bound one."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
if "--late-env" in sys.argv:                 # after libamdhip64 is loaded, before the first HIP call: does the runtime still read it?
    os.environ["GPU_MAX_HW_QUEUES"] = "8"
from bench import seeded_scalars
from fourq_amd import Engine, codec, constants

lg = int(sys.argv[1]) if len(sys.argv) > 1 else 20
LINK = "--no-link" not in sys.argv
STREAMS = "--streams" in sys.argv          # only use two torch streams for a trivial kernel each, no copies
n = 1 << lg
dev = torch.device("cuda", 0)
def link_test():
    """64 MiB each way, alone and together (torch copies on two torch streams)"""
    a_h = torch.empty(64 << 20, dtype=torch.uint8).pin_memory(); b_h = torch.empty(64 << 20, dtype=torch.uint8).pin_memory()
    a_d = torch.empty(64 << 20, dtype=torch.uint8, device=dev); b_d = torch.empty(64 << 20, dtype=torch.uint8, device=dev)
    s1, s2 = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
    def timed(fn, reps=5):
        best = 1e9
        for _ in range(reps):
            torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
        return best
    def h2d():
        with torch.cuda.stream(s1): a_d.copy_(a_h, non_blocking=True)
    def d2h():
        with torch.cuda.stream(s2): b_h.copy_(b_d, non_blocking=True)
    def both():
        h2d(); d2h()
    for name, fn, mb in (("h2d alone", h2d, 64), ("d2h alone", d2h, 64), ("both at once", both, 128)):
        t = timed(fn)
        print("link %-13s %.3f ms  %.1f GB/s" % (name, t * 1e3, mb * 1.048576e-3 / t), flush=True)


if STREAMS:
    x = torch.zeros(16, device=dev)
    for st in (torch.cuda.Stream(dev), torch.cuda.Stream(dev)):
        with torch.cuda.stream(st):
            x.add_(1)
    torch.cuda.synchronize()
elif LINK:
    link_test()
g1 = codec.pack_point((constants.Gx, constants.Gy, (1, 0), constants.Gx, constants.Gy))
with Engine(0) as eng:
    te = eng.table_endo(g1)
    s = eng.host_array(seeded_scalars(1, n))
    pts = eng.host_array(eng.mul_endo_fixed(seeded_scalars(2, n), te))
    out = eng.host_empty((n, 20))
    for name, fn in (("R1", lambda: eng.mul_endo(s, pts, out=out)),):
        fn()
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter(); fn(); best = min(best, time.perf_counter() - t0)
        st = eng.host_stats()
        print("%-6s n=2^%d: %.3f ms -> %.1f Mmults/s (chunks %d, copies %.1f / %.1f GB/s)" % (name, lg, best * 1e3, n / best / 1e6, st["chunks"], st["gbs_h2d"] or 0, st["gbs_d2h"] or 0), flush=True)
