#!/usr/bin/env python3
"""Runs MUL_endo through prep_kernel + ladder_kernel<PREBUILT> at 2^16 (FOURQ_SPLIT_ALL) a few times: a target for
rocprofv3 --pmc when the table construction and the ladder are to be looked at separately (GPU box)."""
import os
os.environ.setdefault("FOURQ_DEBUG_ROUTES", "1")      # the FOURQ_* route hooks below are read only under this gate (tools/README.md)
import sys

os.environ["FOURQ_SPLIT_ALL"] = "1"
os.environ["FOURQ_SPLIT_MIN"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from bench import seeded_scalars
from fourq_amd import Engine, codec, constants

dev = torch.device("cuda", 0)
eng = Engine(0)
n = 1 << 16
g1 = codec.pack_point((constants.Gx, constants.Gy, (1, 0), constants.Gx, constants.Gy))
s = torch.from_numpy(seeded_scalars(1, n).view(np.int64)).to(dev)
k = torch.from_numpy(seeded_scalars(2, n).view(np.int64)).to(dev)
pts = torch.empty((n, 20), dtype=torch.int64, device=dev)
out = torch.empty((n, 20), dtype=torch.int64, device=dev)
eng.mul_endo_fixed_dev(k, eng.table_endo(g1), pts, n)
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 60):
    eng.mul_endo_dev(s, pts, out, n)
eng.sync()
