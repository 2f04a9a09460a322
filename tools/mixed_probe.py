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
te = eng.table_endo(g1)
n = int(os.environ.get('MIXED_N', str(1 << 17)))
print('n', n, 'queue', os.environ.get('FOURQ_MIXED_QUEUE'), 'ct', os.environ.get('FOURQ_CT_SELECT'))
s = torch.from_numpy(seeded_scalars(1, n).view(np.int64)).to(dev)
k = torch.from_numpy(seeded_scalars(2, n).view(np.int64)).to(dev)
pts = torch.empty((n, 20), dtype=torch.int64, device=dev); eng.mul_endo_fixed_dev(k, te, pts, n)
out = torch.empty((n, 20), dtype=torch.int64, device=dev)
for name, fl in (("all fixed", np.zeros(n, np.uint8)), ("all var", np.ones(n, np.uint8)), ("50/50", (np.arange(n) & 1).astype(np.uint8))):
    f = torch.from_numpy(fl).to(dev)
    for _ in range(3): eng.mul_endo_mixed_dev(s, pts, f, te, out, n)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(stream); eng.mul_endo_mixed_dev(s, pts, f, te, out, n); b.record(stream); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b))
    print(name, "%.3f ms" % best)
for name, fn in (("plain fixed 2^16", lambda: eng.mul_endo_fixed_dev(s, te, out, n // 2)), ("plain var 2^16", lambda: eng.mul_endo_dev(s, pts, out, n // 2)),
                 ("plain fixed 2^17", lambda: eng.mul_endo_fixed_dev(s, te, out, n)), ("plain var 2^17", lambda: eng.mul_endo_dev(s, pts, out, n))):
    fn(); torch.cuda.synchronize(); best = 1e9
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(stream); fn(); b.record(stream); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b))
    print(name, "%.3f ms" % best)
