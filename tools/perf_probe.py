#!/usr/bin/env python3
"""Throughput probe for kernel experiments (GPU box): mults/s by kernel-event time for several
modes and batch sizes.   python tools/perf_probe.py [--modes endo_var,endo_fixed,...] [--sizes 16,18,20]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from bench import seeded_scalars
from fourq_amd import Engine, codec, constants

ap = argparse.ArgumentParser()
ap.add_argument("--modes", default="endo_var,endo_fixed,win_var,win_fixed,dh_endo")
ap.add_argument("--sizes", default="16,18,20", help="log2 batch sizes")
ap.add_argument("--ns", default="", help="explicit batch sizes (overrides --sizes)")
ap.add_argument("--reps", type=int, default=5)
args = ap.parse_args()

dev = torch.device("cuda", 0)
stream = torch.cuda.Stream(device=dev)
torch.cuda.set_stream(stream)
eng = Engine(0, stream=stream.cuda_stream)
g1 = codec.pack_point((constants.Gx, constants.Gy, (1, 0), constants.Gx, constants.Gy))
te, tw = eng.table_endo(g1), eng.table_windowed(g1)
print("lib", os.environ.get("FOURQ_AMD_LIB", "default"), "lanes", eng.lanes, flush=True)
for n in ([int(x) for x in args.ns.split(",")] if args.ns else [1 << int(x) for x in args.sizes.split(",")]):
    lg = n.bit_length() - 1
    s = torch.from_numpy(seeded_scalars(1, n).view(np.int64)).to(dev)
    k = torch.from_numpy(seeded_scalars(2, n).view(np.int64)).to(dev)
    pts = torch.empty((n, 20), dtype=torch.int64, device=dev)
    eng.mul_endo_fixed_dev(k, te, pts, n)
    out = torch.empty((n, 20), dtype=torch.int64, device=dev)
    aff = torch.empty((n, 8), dtype=torch.int64, device=dev)
    aff_out = torch.empty((n, 8), dtype=torch.int64, device=dev)
    st = torch.empty(n, dtype=torch.uint8, device=dev)
    eng.dh_endo_dev(k, torch.from_numpy(np.repeat(codec.pack_point((constants.Gx, constants.Gy)).reshape(1, 8), n, 0).view(np.int64)).to(dev), te, aff, st, n)
    torch.cuda.synchronize()
    g392 = eng.mul_endo(codec.pack_scalars([392]), g1.reshape(1, 20))[0]
    comb = eng.comb_table(g392)
    t392 = eng.table_endo(g392)
    gaff = torch.from_numpy(np.repeat(codec.pack_point((constants.Gx, constants.Gy)).reshape(1, 8), n, 0).view(np.int64)).to(dev)
    fns = {
        "comb": lambda: eng.comb_mul_dev(s, comb, aff_out, st, n),
        "dh_fixed": lambda: eng.dh_endo_dev(s, gaff, t392, aff_out, st, n),
        "endo_var": lambda: eng.mul_endo_dev(s, pts, out, n),
        "endo_fixed": lambda: eng.mul_endo_fixed_dev(s, te, out, n),
        "win_var": lambda: eng.mul_windowed_dev(s, pts, out, n),
        "win_fixed": lambda: eng.mul_windowed_fixed_dev(s, tw, out, n),
        "dh_endo": lambda: eng.dh_endo_dev(s, aff, None, aff_out, st, n),
    }
    for mode in args.modes.split(","):
        fn = fns[mode]
        fn(); torch.cuda.synchronize()
        best = 1e9
        for _ in range(args.reps):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(stream); fn(); b.record(stream); torch.cuda.synchronize()
            best = min(best, a.elapsed_time(b))
        print("n=%-8d (2^%d+%d) %-10s %8.3f ms  %8.2f Mmults/s" % (n, lg, n - (1 << lg), mode, best, n / best / 1e3), flush=True)
