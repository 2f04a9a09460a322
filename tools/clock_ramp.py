#!/usr/bin/env python3
"""How long the GPU takes to reach its steady clock under the headline kernel (GPU box): per-launch
kernel time of cfg2 by HIP events, from a cold start.   python tools/clock_ramp.py [launches]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from bench import seeded_scalars
from fourq_amd import Engine, codec, constants

launches = int(sys.argv[1]) if len(sys.argv) > 1 else 600
dev = torch.device("cuda", 0)
stream = torch.cuda.Stream(device=dev)
torch.cuda.set_stream(stream)
eng = Engine(0, stream=stream.cuda_stream)
n = 1 << 16
g1 = codec.pack_point((constants.Gx, constants.Gy, (1, 0), constants.Gx, constants.Gy))
s = torch.from_numpy(seeded_scalars(1, n).view(np.int64)).to(dev)
k = torch.from_numpy(seeded_scalars(2, n).view(np.int64)).to(dev)
pts = torch.empty((n, 20), dtype=torch.int64, device=dev)
out = torch.empty((n, 20), dtype=torch.int64, device=dev)
eng.mul_endo_fixed_dev(k, eng.table_endo(g1), pts, n)
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(launches + 1)]
ev[0].record(stream)
for i in range(launches):
    eng.mul_endo_dev(s, pts, out, n)
    ev[i + 1].record(stream)
torch.cuda.synchronize()
t = [ev[i].elapsed_time(ev[i + 1]) for i in range(launches)]
for lo in range(0, launches, 25):
    chunk = t[lo:lo + 25]
    print("launch %4d..%4d  mean %.4f ms  min %.4f  max %.4f   (t = %.1f ms)" % (lo, lo + len(chunk) - 1, sum(chunk) / len(chunk), min(chunk), max(chunk), sum(t[:lo])))
