#!/usr/bin/env python3
"""Would TWO kernel streams, alternating over the chunks of a host-array call, remove the ~22 us the kernel stream idles per chunk
?  The fused ladder takes a CU's whole LDS, so a kernel on the second stream cannot start a workgroup before the
first stream's kernel has retired one: the two run back to back by construction, but the second one's launch, its cross-stream wait and the
event record behind the first are processed WHILE the first computes.  This probe emulates the kernel side of the pipeline with two
engines (two contexts = two streams + two scratch areas): 16 generations, each launch preceded by a wait on an already-recorded event of a
third stream and followed by an event record, (a) all on one stream, (b) alternating between the two.  No copies."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from bench import seeded_scalars
from fourq_amd import Engine, codec, constants

dev = torch.device("cuda", 0)
g1 = codec.pack_point((constants.Gx, constants.Gy, (1, 0), constants.Gx, constants.Gy))
GENS = 16
sA, sB, sC = torch.cuda.Stream(dev), torch.cuda.Stream(dev), torch.cuda.Stream(dev)
with Engine(0, stream=sA.cuda_stream) as ea, Engine(0, stream=sB.cuda_stream) as eb:
    n = ea.lanes
    te = ea.table_endo(g1)
    s_h = seeded_scalars(1, n)
    p_h = ea.mul_endo_fixed(seeded_scalars(2, n), te)
    s, p = (torch.from_numpy(a.view(np.int64)).to(dev) for a in (s_h, p_h))
    outs = [torch.empty((n, 20), dtype=torch.int64, device=dev) for _ in range(GENS)]
    torch.cuda.synchronize()

    def run(mode, events):
        """mode 1: every launch on stream A; mode 2: alternating A / B.  events: a wait on a signalled event of stream C in front of every
        launch and an event record behind it (what a pipeline chunk has on the kernel stream)."""
        evs = [torch.cuda.Event() for _ in range(GENS)]
        done = [torch.cuda.Event() for _ in range(GENS)]
        for e in evs:
            e.record(sC)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for g in range(GENS):
            eng, st = (ea, sA) if (mode == 1 or g % 2 == 0) else (eb, sB)
            if events:
                st.wait_event(evs[g])
            eng.mul_endo_dev(s, p, outs[g], n)
            if events:
                done[g].record(st)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) * 1e3

    for _ in range(30):
        ea.mul_endo_dev(s, p, outs[0], n); eb.mul_endo_dev(s, p, outs[1], n)
    torch.cuda.synchronize()
    for label, mode, events in (("one stream, bare launches", 1, False), ("one stream, wait + record per launch", 1, True),
                                ("two streams alternating, bare", 2, False), ("two streams alternating, wait + record", 2, True)):
        ts = sorted(run(mode, events) for _ in range(15))
        print("%-42s %d generations: best %.3f ms  median %.3f ms" % (label, GENS, ts[0], ts[len(ts) // 2]), flush=True)
    want = outs[0].cpu()
    assert all(torch.equal(o.cpu(), want) for o in outs), "outputs differ between generations"
    print("all %d outputs identical" % GENS)
