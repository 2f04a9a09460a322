#!/usr/bin/env python3
"""Config 5's 7 %: is it the one wave that must run two variable-base items?  A mixed batch of 2^17 elements, the number
of variable-base elements stepped across 1 024 x 64 = 65 536 (config 5's seeded flags give 65 550): back-to-back steps at the sustained
clock, HIP events on the launch stream, the in-kernel clock beside them, cycles per step = time x clock."""
import os, sys
os.environ.setdefault("FOURQ_DEBUG_ROUTES", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench import seeded_scalars, seeded_flags
from fourq_amd import Engine, codec, constants
dev = torch.device("cuda", 0)
stream = torch.cuda.Stream(device=dev); torch.cuda.set_stream(stream)
eng = Engine(0, stream=stream.cuda_stream)
g1 = codec.pack_point((constants.Gx, constants.Gy, (1, 0), constants.Gx, constants.Gy))
te = eng.table_endo(g1)
n = 1 << 17
s = torch.from_numpy(seeded_scalars(50004, n).view(np.int64)).to(dev)
k = torch.from_numpy(seeded_scalars(50003, n).view(np.int64)).to(dev)
pts = torch.empty((n, 20), dtype=torch.int64, device=dev); eng.mul_endo_fixed_dev(k, te, pts, n)
out = torch.empty((n, 20), dtype=torch.int64, device=dev)
base = seeded_flags(50002, n)
print("config 5's own flags: %d variable-base, %d fixed-base" % (int(base.sum()), int(n - base.sum())))
def with_count(n_var):
    """config 5's flags with the first surplus variable-base flags cleared (or fixed-base ones set) until exactly n_var are set"""
    f = base.copy()
    have = int(f.sum())
    idx = np.flatnonzero(f == (1 if have > n_var else 0))[: abs(have - n_var)]
    f[idx] ^= 1
    assert int(f.sum()) == n_var
    return f
for n_var in (65472, 65522, 65536, 65537, 65550, 65600, 66000):
    f = torch.from_numpy(with_count(n_var)).to(dev)
    for _ in range(150): eng.mul_endo_mixed_dev(s, pts, f, te, out, n)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    steps = 300
    a.record(stream)
    for _ in range(steps): eng.mul_endo_mixed_dev(s, pts, f, te, out, n)
    b.record(stream)
    for _ in range(200): eng.mul_endo_mixed_dev(s, pts, f, te, out, n)
    clk = eng.diag_clock(25000)
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / steps
    print("n_var %6d (%4d whole items + %2d lanes): %.4f ms per step at %.0f MHz = %.0f k cycles" % (n_var, n_var // 64, n_var % 64, ms, clk["mhz"], ms * clk["mhz"]), flush=True)
eng.close()
