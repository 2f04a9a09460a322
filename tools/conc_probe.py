import os, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from bench import seeded_scalars
from fourq_amd import Engine, codec, constants
dev = torch.device("cuda", 0)
g1 = codec.pack_point((constants.Gx, constants.Gy, (1, 0), constants.Gx, constants.Gy))
K = 4
engs = [Engine(0) for _ in range(K)]
te = engs[0].table_endo(g1)
n = 16384
s = [torch.from_numpy(seeded_scalars(1 + k, n).view(np.int64)).to(dev) for k in range(K)]
p = [torch.empty((n, 20), dtype=torch.int64, device=dev) for _ in range(K)]
o = [torch.empty((n, 20), dtype=torch.int64, device=dev) for _ in range(K)]
for k in range(K):
    engs[k].mul_endo_fixed_dev(s[k], te, p[k], n); engs[k].sync()
def run(which):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        for k in which: engs[k].mul_endo_dev(s[k], p[k], o[k], n)
    for k in which: engs[k].sync()
    return (time.perf_counter() - t0) / 20 * 1e3
for _ in range(3): run(range(K))
print("one engine, one 16384 batch per step: %.3f ms" % run([0]))
print("four engines (four streams), one 16384 batch each per step: %.3f ms" % run(range(K)))
