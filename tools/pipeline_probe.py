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

lg = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 20
FORMATS = next((a.split("=", 1)[1].split(",") for a in sys.argv if a.startswith("--formats=")), ["r1"])     # r1, affine, bytes, fixed (cfg3's call)
PAGEABLE = "--pageable" in sys.argv          # plain numpy arrays in and out (the bounce path) instead of pinned ones
TIMING = "--timing" in sys.argv              # fourq_ctx_set_host_timing: the four event records per chunk that rounds 2-4 always made
FLOOR = "--floor" in sys.argv                # also time each format device-resident (one _dev call for the whole batch): the kernels' own pace
REPS = next((int(a.split("=", 1)[1]) for a in sys.argv if a.startswith("--reps=")), 3)
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
ON_TORCH_STREAM = "--torch-stream" in sys.argv      # the engine enqueues on a torch side stream, as bench.py's does
SECOND_CTX = any(a.startswith("--second-ctx") for a in sys.argv)             # a second context on the same stream that has run kernels (bench.py's constant-time engine)
side = torch.cuda.Stream(dev) if ON_TORCH_STREAM else None
if side is not None:
    torch.cuda.set_stream(side)
other = None
if SECOND_CTX:
    other = Engine(0, stream=side.cuda_stream if side is not None else None)
    t2 = other.table_endo(g1)
    other.mul_endo_fixed(seeded_scalars(9, 4096), t2)
    if "--second-ctx-host" in sys.argv:             # ... and that has made a pipelined host-array call of its own (its copy streams have been used)
        other.mul_endo_fixed(seeded_scalars(9, 4 << 16), t2)
print("GPU_MAX_HW_QUEUES=%s torch_stream=%s second_ctx=%s" % (os.environ.get("GPU_MAX_HW_QUEUES"), ON_TORCH_STREAM, SECOND_CTX), flush=True)
with Engine(0, stream=side.cuda_stream if side is not None else None) as eng:
    if PAGEABLE:
        eng.host_array = lambda a, dtype=None: np.ascontiguousarray(a)
        eng.host_empty = lambda shape, dtype=np.uint64: np.empty(shape, dtype=dtype)
    te = eng.table_endo(g1)
    s = eng.host_array(seeded_scalars(1, n))
    eng.host_timing(TIMING)
    calls, floors = {}, {}
    def dev(a):
        a = np.ascontiguousarray(a)
        return torch.from_numpy(a.view(np.int64) if a.dtype == np.uint64 else a).to("cuda:0")
    s_d = dev(np.asarray(s)) if FLOOR else None
    if "r1" in FORMATS:
        pts = eng.host_array(eng.mul_endo_fixed(seeded_scalars(2, n), te))
        out = eng.host_empty((n, 20))
        calls["r1"] = lambda: eng.mul_endo(s, pts, out=out)
        if FLOOR:
            p_d, o_d = dev(np.asarray(pts)), torch.empty((n, 20), dtype=torch.int64, device="cuda:0")
            floors["r1"] = lambda: eng.mul_endo_dev(s_d, p_d, o_d, n)
    if "fixed" in FORMATS:
        tw = eng.table_windowed(g1)
        outf = eng.host_empty((n, 20))
        calls["fixed"] = lambda: eng.mul_windowed_fixed(s, tw, out=outf)
        if FLOOR:
            of_d = torch.empty((n, 20), dtype=torch.int64, device="cuda:0")
            floors["fixed"] = lambda: eng.mul_windowed_fixed_dev(s_d, tw, of_d, n)
    if "affine" in FORMATS or "bytes" in FORMATS:
        g_aff = np.repeat(codec.pack_point((constants.Gx, constants.Gy)).reshape(1, 8), n, axis=0)
        aff_h, status = eng.dh_endo(seeded_scalars(2, n), g_aff)             # canonical affine N-torsion points
        assert not status.any()
        aff = eng.host_array(aff_h)
        oa = eng.host_empty((n, 8))
        calls["affine"] = lambda: eng.mul_affine(s, aff, out=oa)
        if FLOOR:
            a_d, oa_d = dev(aff_h), torch.empty((n, 8), dtype=torch.int64, device="cuda:0")
            eng.reserve(n)
            floors["affine"] = lambda: eng.mul_affine_dev(s_d, a_d, oa_d, n)
        if "bytes" in FORMATS:
            enc = eng.host_array(eng.encode(aff_h))
            oe, ost = eng.host_empty((n, 32), np.uint8), eng.host_empty((n,), np.uint8)
            calls["bytes"] = lambda: eng.mul_bytes(s, enc, out=oe, status=ost)
            if FLOOR:
                e_d, oe_d, st_d = dev(np.asarray(enc)), torch.empty((n, 32), dtype=torch.uint8, device="cuda:0"), torch.empty(n, dtype=torch.uint8, device="cuda:0")
                floors["bytes"] = lambda: eng.mul_bytes_dev(s_d, e_d, oe_d, st_d, n)
    for name in FORMATS:
        fn = calls[name]
        fn()
        times = []
        for _ in range(REPS):
            t0 = time.perf_counter(); fn(); times.append(time.perf_counter() - t0)
        best, med = min(times), sorted(times)[len(times) // 2]
        st = eng.host_stats()
        floor = ""
        if name in floors:
            for _ in range(2):
                floors[name]()
            eng.sync()
            ft = []
            for _ in range(REPS):
                t0 = time.perf_counter(); floors[name](); eng.sync(); ft.append(time.perf_counter() - t0)
            floor = "  device-resident %.3f ms" % (min(ft) * 1e3)
        kern = "  kernel stream: busy %.3f ms in a span of %.3f" % (st["kernels_ms"], st["kernels_span_ms"]) if st.get("kernels_ms") else ""
        print("%-6s n=2^%d: best %.3f ms  median %.3f ms -> %.1f Mmults/s (chunks %d, copies %.1f / %.1f GB/s)%s%s" % (
            name, lg, best * 1e3, med * 1e3, n / best / 1e6, st["chunks"], st["gbs_h2d"] or 0, st["gbs_d2h"] or 0, kern, floor), flush=True)
