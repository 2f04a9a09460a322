#!/usr/bin/env python3
"""Fused launch vs prep + ladder for variable-base MUL_endo at batch sizes that do not fill whole generations
of resident lanes (GPU box).   python tools/route_probe.py [n ...]"""
import os
os.environ.setdefault("FOURQ_DEBUG_ROUTES", "1")      # the FOURQ_* route hooks below are read only under this gate (tools/README.md)
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from bench import seeded_scalars
from fourq_amd import Engine, codec, constants

sizes = [int(x) for x in sys.argv[1:]] or [65536, 66000, 70000, 80000, 98304, 120000, 131072, 140000, 196608, 200000, 262144]
dev = torch.device("cuda", 0)
stream = torch.cuda.Stream(device=dev)
torch.cuda.set_stream(stream)
engines = {}
for name, env in (("fused", {"FOURQ_SPLIT_MIN": str(1 << 31)}), ("split", {"FOURQ_SPLIT_ALL": "1", "FOURQ_SPLIT_MIN": "1"}), ("auto", {})):
    for k in ("FOURQ_SPLIT_ALL", "FOURQ_SPLIT_MIN"):
        os.environ.pop(k, None)
    os.environ.update(env)
    engines[name] = Engine(0, stream=stream.cuda_stream)
g1 = codec.pack_point((constants.Gx, constants.Gy, (1, 0), constants.Gx, constants.Gy))
nmax = max(sizes)
s = torch.from_numpy(seeded_scalars(1, nmax).view(np.int64)).to(dev)
k = torch.from_numpy(seeded_scalars(2, nmax).view(np.int64)).to(dev)
pts = torch.empty((nmax, 20), dtype=torch.int64, device=dev)
out = torch.empty((nmax, 20), dtype=torch.int64, device=dev)
e0 = engines["fused"]
e0.mul_endo_fixed_dev(k, e0.table_endo(g1), pts, nmax)
for _ in range(100):                                  # bring the clock up (tools/clock_ramp.py)
    e0.mul_endo_dev(s, pts, out, 65536)
torch.cuda.synchronize()
for n in sizes:
    row = []
    for name, e in engines.items():
        best = 1e9
        for _ in range(5):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(stream); e.mul_endo_dev(s, pts, out, n); b.record(stream); torch.cuda.synchronize()
            best = min(best, a.elapsed_time(b))
        row.append("%s %.3f ms" % (name, best))
    print("n=%7d  %s" % (n, "   ".join(row)), flush=True)
