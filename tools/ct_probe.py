#!/usr/bin/env python3
"""Counter-invariance probe for the table-selection modes (GPU box; run under rocprofv3 by tools/ct_invariance.sh).

    python3 tools/ct_probe.py --mode ct|default --scalars random|zero|ones|same --n 65536 [--reps 4]

Launches the hot kernels of every configuration (variable-base MUL_endo, fixed-base MUL_endo / MUL_windowed, comb keygen,
DH_endo) on ONE class of secret scalars.  What draft-ladd-cfrg-4q.md:753-758 demands of a constant-time implementation
is that "memory addresses accessed do not depend on secret data": with FOURQ_CT_SELECT every counter that reflects
addresses -- bytes fetched and written at the L2's memory side, LDS bank-conflict cycles, instruction counts, duration --
must be the same whatever the scalars are, while the default mode (a digit of the scalar is a table address, as in the
reference) is expected to show its dependence exactly there.  Scalar classes:
    random   independent uniform 256-bit scalars (every lane its own digits)
    zero     m = 0 in every lane (all lanes the same digit string)
    ones     m = 2^256 - 1 in every lane
    same     one random scalar replicated in every lane
The points are the same in every class (public data).
"""
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
ap.add_argument("--mode", choices=["ct", "default"], required=True)
ap.add_argument("--scalars", choices=["random", "zero", "ones", "same"], required=True)
ap.add_argument("--n", type=int, default=1 << 16)
ap.add_argument("--reps", type=int, default=4)
args = ap.parse_args()

dev = torch.device("cuda", 0)
stream = torch.cuda.Stream(device=dev)
torch.cuda.set_stream(stream)
eng = Engine(0, stream=stream.cuda_stream)
n = args.n
g1 = codec.pack_point((constants.Gx, constants.Gy, (1, 0), constants.Gx, constants.Gy))
gaff = codec.pack_point((constants.Gx, constants.Gy))
te, tw = eng.table_endo(g1), eng.table_windowed(g1)
g392 = eng.mul_endo(codec.pack_scalars([392]), g1.reshape(1, 20))[0]
comb = eng.comb_table(g392)

# public inputs, identical for every scalar class: projective points and affine public keys from fixed seeds
k = torch.from_numpy(seeded_scalars(2, n).view(np.int64)).to(dev)
pts = torch.empty((n, 20), dtype=torch.int64, device=dev)
eng.mul_endo_fixed_dev(k, te, pts, n)
pub = torch.empty((n, 8), dtype=torch.int64, device=dev)
st = torch.empty(n, dtype=torch.uint8, device=dev)
eng.comb_mul_dev(k, comb, pub, st, n)
torch.cuda.synchronize()

if args.scalars == "random":
    s_h = seeded_scalars(1, n)
elif args.scalars == "zero":
    s_h = np.zeros((n, 4), dtype=np.uint64)
elif args.scalars == "ones":
    s_h = np.full((n, 4), np.uint64(0xFFFFFFFFFFFFFFFF), dtype=np.uint64)
else:
    s_h = np.repeat(seeded_scalars(7, 1), n, axis=0)
s = torch.from_numpy(s_h.view(np.int64)).to(dev)
out = torch.empty((n, 20), dtype=torch.int64, device=dev)
aff = torch.empty((n, 8), dtype=torch.int64, device=dev)

eng.ct_select = args.mode == "ct"           # everything above ran in the default mode: only the launches below are the subject
torch.cuda.synchronize()
print("PROBE mode=%s scalars=%s n=%d" % (args.mode, args.scalars, n), flush=True)
for _ in range(args.reps):
    eng.mul_endo_dev(s, pts, out, n)
    eng.mul_endo_fixed_dev(s, te, out, n)
    eng.mul_windowed_fixed_dev(s, tw, out, n)
    eng.comb_mul_dev(s, comb, aff, st, n)
    eng.dh_endo_dev(s, pub, None, aff, st, n)
torch.cuda.synchronize()
eng.close()
