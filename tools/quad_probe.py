#!/usr/bin/env python3
"""Kernel time of small variable-base and fixed-base batches with one, two and four lanes per element (GPU box).
    python tools/quad_probe.py        -> table, ms per call (device-resident, HIP events, best of 7)"""
import os, subprocess, sys
os.environ.setdefault("FOURQ_DEBUG_ROUTES", "1")      # the FOURQ_* route hooks below are read only under this gate (tools/README.md)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

def child():
    import numpy as np, torch
    from bench import seeded_scalars
    from fourq_amd import Engine, codec, constants
    dev = torch.device("cuda", 0)
    stream = torch.cuda.Stream(device=dev); torch.cuda.set_stream(stream)
    eng = Engine(0, stream=stream.cuda_stream)
    eng.ct_select = os.environ.get("PROBE_CT") == "1"
    g1 = codec.pack_point((constants.Gx, constants.Gy, (1, 0), constants.Gx, constants.Gy))
    te, tw = eng.table_endo(g1), eng.table_windowed(g1)
    nmax = 32768
    s = torch.from_numpy(seeded_scalars(1, nmax).view(np.int64)).to(dev)
    k = torch.from_numpy(seeded_scalars(2, nmax).view(np.int64)).to(dev)
    pts = torch.empty((nmax, 20), dtype=torch.int64, device=dev); eng.mul_endo_fixed_dev(k, te, pts, nmax)
    aff = torch.from_numpy(np.repeat(codec.pack_point((constants.Gx, constants.Gy)).reshape(1, 8), nmax, axis=0).view(np.int64)).to(dev)
    out = torch.empty((nmax, 20), dtype=torch.int64, device=dev); st = torch.empty(nmax, dtype=torch.uint8, device=dev)
    def t(fn):
        for _ in range(3): fn()
        torch.cuda.synchronize(); best = 1e9
        for _ in range(7):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(stream); fn(); b.record(stream); torch.cuda.synchronize()
            best = min(best, a.elapsed_time(b))
        return best
    for _ in range(200): eng.mul_endo_dev(s, pts, out, 16384)      # clock
    row = []
    for n in (1, 1024, 16384, 32768):
        row.append("%d: endo %.3f win %.3f dh %.3f fix_e %.3f fix_w %.3f" % (n, t(lambda: eng.mul_endo_dev(s, pts, out, n)), t(lambda: eng.mul_windowed_dev(s, pts, out, n)),
                   t(lambda: eng.dh_endo_dev(s, aff, None, out, st, n)), t(lambda: eng.mul_endo_fixed_dev(s, te, out, n)), t(lambda: eng.mul_windowed_fixed_dev(s, tw, out, n))))
    print(os.environ.get("PROBE_NAME"), " | ".join(row), flush=True)

if __name__ == "__main__":
    if os.environ.get("PROBE_NAME"):
        child()
    else:
        for ct in ("0", "1"):
            print("## constant-time selection" if ct == "1" else "## default selection")
            for name, env in (("one lane ", {"FOURQ_PAIR_MAX": "0"}), ("two lanes", {"FOURQ_QUAD_MAX": "0"}), ("four/two ", {})):
                e = dict(os.environ, PROBE_NAME=name, PROBE_CT=ct, **env)
                subprocess.run([sys.executable, os.path.abspath(__file__)], env=e, check=True)
