import os, sys
os.environ.setdefault("FOURQ_DEBUG_ROUTES", "1")      # the FOURQ_* route hooks below are read only under this gate (tools/README.md)
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from bench import seeded_scalars
from fourq_amd import Engine, codec, constants
dev = torch.device("cuda", 0)
stream = torch.cuda.Stream(device=dev); torch.cuda.set_stream(stream)
eng = Engine(0, stream=stream.cuda_stream)
g1 = codec.pack_point((constants.Gx, constants.Gy, (1, 0), constants.Gx, constants.Gy))
comb = eng.comb_table(g1); eng.comb_stage(comb)
nmax = 65536
s = torch.from_numpy(seeded_scalars(1, nmax).view(np.int64)).to(dev)
out = torch.empty((nmax, 8), dtype=torch.int64, device=dev); st = torch.empty(nmax, dtype=torch.uint8, device=dev)
for _ in range(300): eng.comb_mul_dev(s, None, out, st, 16384)
row = []
for n in (1, 1024, 4096, 16384, 32768, 65536):
    best = 1e9
    for _ in range(7):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(stream); eng.comb_mul_dev(s, None, out, st, n); b.record(stream); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b))
    row.append("%d: %.3f" % (n, best))
print(os.environ.get("FOURQ_QUAD_MAX", "default"), " | ".join(row))
