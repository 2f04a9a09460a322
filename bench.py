#!/usr/bin/env python3
"""bench.py -- FourQ scalar-mults/sec, BASELINE.json config 2 per GPU.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

One step = one pass of the hot path over one batch: 2^16 variable-base MUL_endo(m_i, P_i) per GPU
(random 256-bit scalars; P_i = projective N-torsion points, raw R1 outputs of fixed-base [k_i]G),
inputs and outputs resident in HBM.  Ranks are independent (weak scaling, no data-path
collective).  Rank 0 prints ONE JSON line.  Synthetic data; integer arithmetic (dtype "u32x5"
limbs of GF(2^127-1), reported as "u64" words at the ABI).
"""
import argparse
import json
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

BATCH = 1 << 16                      # BASELINE.json configs[1]
BYTES_PER_OP = 32 + 160 + 160        # scalar + R1 in + R1 out
MADS_PER_OP = 100_100                # v_mad_u64_u32 issued per variable-base MUL_endo by this implementation
HBM_PEAK_GBS = 8000.0                # MI355X_MICROARCH.md
VALU_MAD_PEAK = 1024 * 64 / 4.0 * 2.4e9    # measured: a wave64 v_mad_u64_u32 occupies a SIMD for 4 cycles (profiles/true_rates_r01.txt)


def seeded_scalars(seed, n):
    import numpy as np
    rng = random.Random(seed)
    return np.frombuffer(rng.getrandbits(256 * n).to_bytes(32 * n, "little"), dtype="<u8").reshape(n, 4).copy()


def host_cores():
    """Cores this process may really use: scheduler affinity capped by the cgroup CPU quota (the GPU
    box shows every core of the host but grants a 16-core share per GPU)."""
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as fh:
                fields = fh.read().split()
            if path.endswith("cpu.max"):
                if fields[0] != "max":
                    cores = min(cores, max(1, int(fields[0]) // int(fields[1])))
            else:
                quota = int(fields[0])
                with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as fh:
                    period = int(fh.read())
                if quota > 0:
                    cores = min(cores, max(1, quota // period))
            break
        except (OSError, ValueError, IndexError):
            continue
    return min(cores, int(os.environ.get("FOURQ_BENCH_CORES", "16")))


def _cpu_worker(args):
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import curve4q_oracle as o
    pairs, = args
    t0 = time.perf_counter()
    outs = [o.MUL_endo(m, P) for m, P in pairs]
    return time.perf_counter() - t0, outs


def cpu_baseline(scalars, points, gpu_out, target_seconds=12.0):
    """Times the pure-Python oracle (oracle/curve4q_oracle.py, kind "port") on the host cores over a
    bounded sample of the SAME workload, and uses the sample as a parity gate on the GPU result."""
    import multiprocessing as mp
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import curve4q_oracle as o
    from fourq_amd import codec
    cores = host_cores()
    ms, ps = codec.unpack_scalars(scalars[:4]), codec.unpack_points(points[:4])
    t0 = time.perf_counter()
    for m, P in zip(ms, ps):
        o.MUL_endo(m, P)
    per_op = (time.perf_counter() - t0) / 4
    per_core = max(8, min(len(scalars) // cores, int(target_seconds / per_op)))
    total = per_core * cores
    ms, ps = codec.unpack_scalars(scalars[:total]), codec.unpack_points(points[:total])
    chunks = [([(ms[i], ps[i]) for i in range(c * per_core, (c + 1) * per_core)],) for c in range(cores)]
    t0 = time.perf_counter()
    with mp.get_context("spawn").Pool(cores) as pool:
        results = pool.map(_cpu_worker, chunks)
    wall = time.perf_counter() - t0
    outs = [x for _, chunk in results for x in chunk]
    want = codec.pack_points(outs, 5)
    import numpy as np
    if not np.array_equal(want, gpu_out[:total]):
        raise SystemExit("PARITY FAILURE: GPU MUL_endo differs from the oracle on the cpu_baseline sample")
    busy = max(t for t, _ in results)
    return {"value": round(total / busy, 1), "unit": "scalar-mults/s", "cores": cores, "kind": "port",
            "sample": "%d of the batch's (scalar, point) pairs, MUL_endo via oracle/curve4q_oracle.py (pure Python big ints), "
                      "%d per core on %d processes, %.1f s wall; outputs compared bit-exact with the GPU's" % (total, per_core, cores, wall),
            "per_core": round(per_core / busy, 1)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=BATCH, help="elements per GPU per step (default: BASELINE config 2)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist
    from fourq_amd import Engine, codec, constants
    from fourq_amd.dist import gather_rows, init_process_group

    rank, local_rank, world = init_process_group("nccl")
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d" % (args.gpus, world, args.gpus))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    stream = torch.cuda.Stream(device=dev)     # a real (non-null) stream: the engine launches on it, the events time it
    torch.cuda.set_stream(stream)
    eng = Engine(local_rank, stream=stream.cuda_stream)

    n = args.batch
    # ---- synthetic workload, generated on the GPU: rank r uses seeds 20002+2r / 20003+2r
    g1 = codec.pack_point((constants.Gx, constants.Gy, (1, 0), constants.Gx, constants.Gy))
    table_g = eng.table_endo(g1)
    scalars_h = seeded_scalars(20002 + 2 * rank, n)
    scalars = torch.from_numpy(scalars_h.view(np.int64)).to(dev)
    k_dev = torch.from_numpy(seeded_scalars(20003 + 2 * rank, n).view(np.int64)).to(dev)
    points = torch.empty((n, 20), dtype=torch.int64, device=dev)
    eng.mul_endo_fixed_dev(k_dev, table_g, points, n)          # P_i = [k_i]G, raw R1 (projective, Z != 1)
    out = torch.empty((n, 20), dtype=torch.int64, device=dev)
    torch.cuda.synchronize()

    def step():
        eng.mul_endo_dev(scalars, points, out, n)

    for _ in range(args.warmup):
        step()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for a, b in ev:
        a.record(stream)
        step()
        b.record(stream)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    kernel_ms = sum(a.elapsed_time(b) for a, b in ev) / max(1, len(ev))

    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())

    # ---- plumbing check of the only collective the path has: gather the results once (untimed)
    gathered = gather_rows(out, n * world, dst=0) if world > 1 else out
    if rank == 0:
        assert gathered.shape[0] == n * world
        out_h = out.cpu().numpy().view(np.uint64)
        pts_h = points.cpu().numpy().view(np.uint64)
        total = n * world * args.steps
        value = total / elapsed
        ach_gbs = BYTES_PER_OP * n / (kernel_ms * 1e-3) / 1e9
        line = {
            "metric": "FourQ scalar-mults/sec (batch, whole node)", "value": round(value, 1), "unit": "scalar-mults/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
            "config": {"workload": "BASELINE.json configs[1]: batch of 2^16 variable-base MUL_endo(m,P) per GPU, random 256-bit scalars, "
                                   "projective N-torsion points, raw R1 in/out resident in HBM",
                       "batch_per_gpu": n, "parallelism": "independent shards x%d, no data-path collective" % world},
            "roofline": {"bound": "hbm", "achieved": round(ach_gbs, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(ach_gbs / HBM_PEAK_GBS, 6), "traffic": _pmc_traffic(),
                         "kernel": "ladder_kernel<ENDO, variable base>", "kernel_ms": round(kernel_ms, 4),
                         "algorithmic_bytes_per_launch": BYTES_PER_OP * n,
                         "note": "the path is integer-VALU bound, not HBM bound (SURVEY.md 8d): see valu_roofline"},
            "valu_roofline": {"bound": "valu-int (v_mad_u64_u32 issue)", "achieved": round(MADS_PER_OP * n / (kernel_ms * 1e-3) / 1e12, 3),
                              "peak": round(VALU_MAD_PEAK / 1e12, 3), "unit": "Tmad/s",
                              "frac": round(MADS_PER_OP * n / (kernel_ms * 1e-3) / VALU_MAD_PEAK, 4)},
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(scalars_h, pts_h, out_h)
        print(json.dumps(line), flush=True)
    eng.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def _pmc_traffic():
    """HBM bytes per launch from the committed rocprofv3 --pmc summary (profiles/), or None."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        with open(path) as fh:
            return json.load(fh).get("hbm_bytes_per_launch")
    except (OSError, ValueError):
        return None


if __name__ == "__main__":
    main()
