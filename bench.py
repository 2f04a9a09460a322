#!/usr/bin/env python3
"""bench.py -- FourQ scalar-mults/sec, BASELINE.json config 2 per GPU.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

One step = one pass of the hot path over one batch: 2^16 variable-base MUL_endo(m_i, P_i) per GPU
(random 256-bit scalars; P_i = projective N-torsion points, raw R1 outputs of fixed-base [k_i]G),
inputs and outputs resident in HBM.  `--workload cfg3|cfg4|cfg5` runs the other BASELINE.json
configurations at their per-GPU sizes (the headline the driver records is the default, cfg2).  Ranks are independent (weak scaling, no data-path
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
    kind, items, extra = args
    G = (o.Gx, o.Gy)
    t0 = time.perf_counter()
    if kind == "mul_endo":
        outs = [o.MUL_endo(m, P) for m, P in items]
    elif kind == "mul_windowed_fixed":
        outs = [o.MUL_windowed(m, None_r1(o), table=extra) for m in items]
    elif kind == "dh_exchange":
        outs = [o.DH_endo(a, o.DH_endo(b, G, table=extra)) for a, b in items]
    else:  # mixed: (m, P or None)
        outs = [o.MUL_endo(m, P) if P is not None else o.MUL_endo(m, None_r1(o), table=extra) for m, P in items]
    return time.perf_counter() - t0, outs


def None_r1(o):
    return o.AffineToR1(o.Gx, o.Gy)     # with a table the reference ignores the point (curve4q.py:209, :426)


def cpu_baseline(kind, items, extra, expected, what, target_seconds=12.0):
    """Times the pure-Python oracle (oracle/curve4q_oracle.py, kind "port") on the host cores over a bounded
    sample of the SAME workload, and uses that sample as a parity gate on the GPU result (`expected`: the
    GPU's outputs for the same items, as tuples)."""
    import multiprocessing as mp
    cores = host_cores()
    t0 = time.perf_counter()
    _cpu_worker((kind, items[:4], extra))
    per_op = (time.perf_counter() - t0) / 4
    per_core = max(8, min(len(items) // cores, int(target_seconds / per_op)))
    total = per_core * cores
    chunks = [(kind, items[c * per_core:(c + 1) * per_core], extra) for c in range(cores)]
    t0 = time.perf_counter()
    with mp.get_context("spawn").Pool(cores) as pool:
        results = pool.map(_cpu_worker, chunks)
    wall = time.perf_counter() - t0
    outs = [x for _, chunk in results for x in chunk]
    if outs != expected[:total]:
        raise SystemExit("PARITY FAILURE: GPU result differs from the oracle on the cpu_baseline sample (%s)" % what)
    busy = max(t for t, _ in results)
    return {"value": round(total / busy, 1), "unit": "scalar-mults/s", "cores": cores, "kind": "port",
            "sample": "%d units of the timed batch, %s via oracle/curve4q_oracle.py (pure Python big ints), %d per core on %d "
                      "processes, %.1f s wall; outputs compared bit-exact with the GPU's" % (total, what, per_core, cores, wall),
            "per_core": round(per_core / busy, 1)}


def c_port_baseline(workload, scalars_h, second_h, points_h, table_h, flags_h, expected_words):
    """The C restatement (oracle/fourq_oracle.c, OpenMP) over the WHOLE timed batch: a second parity gate on every
    output and, for context, its rate on the same host cores."""
    os.environ.setdefault("OMP_NUM_THREADS", str(host_cores()))
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import numpy as np
    import oracle_c as oc
    from fourq_amd import codec, constants
    oc.lib()
    t0 = time.perf_counter()
    if workload == "cfg2":
        got = oc.mul(oc.ENDO, scalars_h, points_h)
    elif workload == "cfg3":
        got = oc.mul(oc.WINDOWED, scalars_h, None, table_h)
    elif workload == "cfg4":
        g = np.repeat(codec.pack_point((constants.Gx, constants.Gy)).reshape(1, 8), len(scalars_h), axis=0)
        mid, st1 = oc.dh(oc.ENDO, second_h, g)
        got, st2 = oc.dh(oc.ENDO, scalars_h, mid)
        if st1.any() or st2.any():
            raise SystemExit("cfg4: unexpected DH failure status in the C oracle")
    else:
        fixed = oc.mul(oc.ENDO, scalars_h, None, table_h)
        var = oc.mul(oc.ENDO, scalars_h, points_h)
        got = np.where(flags_h.reshape(-1, 1) != 0, var, fixed)
    dt = time.perf_counter() - t0
    if not np.array_equal(got, expected_words):
        raise SystemExit("PARITY FAILURE: GPU result differs from the C oracle on the full batch (%s)" % workload)
    units = len(scalars_h) * (2 if workload == "cfg5" else 1)     # cfg5 evaluates both variants of every element
    return {"value": round(units / dt, 1), "unit": "scalar-mults/s", "threads": oc.num_threads(), "cores": host_cores(), "kind": "port",
            "sample": "whole batch (%d units) via oracle/fourq_oracle.c, every output compared bit-exact with the GPU's" % len(scalars_h)}


def edge_case_check(eng):
    """every run carries an edge-case mini-batch -- scalars {0,1,2,N-1,N,N+1,2N,2^255,2^256-1}
    on G and -G through MUL_endo, MUL_windowed and DH_endo, bit-exact against the Python oracle (part of the
    cpu_baseline leg: the oracle is the checker).  Returns the number of cases."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import curve4q_oracle as o
    from fourq_amd import codec
    N = o.N
    ms = [0, 1, 2, N - 1, N, N + 1, 2 * N, 1 << 255, (1 << 256) - 1]
    G = (o.Gx, o.Gy)
    negG = (o.GFp2.neg(o.Gx), o.Gy)
    cases = [(m, A) for A in (G, negG) for m in ms]
    s = codec.pack_scalars([m for m, _ in cases])
    r1 = codec.pack_points([o.AffineToR1(*A) for _, A in cases], 5)
    aff = codec.pack_points([A for _, A in cases], 2)
    for name, got, want in (("MUL_endo", eng.mul_endo(s, r1), o.MUL_endo), ("MUL_windowed", eng.mul_windowed(s, r1), o.MUL_windowed)):
        if codec.unpack_points(got) != [want(m, o.AffineToR1(*A)) for m, A in cases]:
            raise SystemExit("PARITY FAILURE: %s on the edge-case mini-batch" % name)
    out, status = eng.dh_endo(s, aff)
    for (m, A), row, st in zip(cases, codec.unpack_points(out), status):
        try:
            want, code = o.DH_endo(m, A), 0
        except Exception as exc:            # the reference's two rejections (curve4q.py:448, :460)
            want, code = None, (1 if "not on curve" in str(exc) else 2)
        if int(st) != code or (code == 0 and row != want):
            raise SystemExit("PARITY FAILURE: DH_endo on the edge-case mini-batch (m=%d)" % m)
    return 3 * len(cases)


# per-workload constants: algorithmic bytes per unit and v_mad_u64_u32 issued per unit by this
# implementation, counted from the formulas: GF(p^2) M = 100, S = 50; DBL = 3M+4S = 500,
# ADD = 8M = 800; ladder step 1 300 (x64 endo, 4 DBL + ADD = 2 800 x62 windowed); table_endo 14 300;
# DH extras (membership, x392, inversion) 8 600; comb 9 DBL + 49 mixed ADD (7M) = 38 800 + inversion 2 500; at cfg4's size
# eight elements share one inversion (normalize_kernel<8>): -1 800 per DH_core
WORKLOADS = {
    "cfg2": dict(batch=1 << 16, bytes=32 + 160 + 160, mads=97_600, kernel="ladder_kernel<ENDO, FUSED>",
                 text="BASELINE.json configs[1]: batch of 2^16 variable-base MUL_endo(m,P) per GPU, random 256-bit scalars, "
                      "projective N-torsion points, raw R1 in/out resident in HBM"),
    "cfg3": dict(batch=1 << 20, bytes=32 + 160, mads=173_600, kernel="ladder_kernel<WINDOWED, LDS>",
                 text="BASELINE.json configs[2]: batch of 2^20 fixed-base MUL_windowed(m,G,table) per GPU, table staged in LDS, raw R1 out"),
    "cfg4": dict(batch=1 << 19, bytes=2 * 161, mads=39_400 + 97_600 + 6_800, kernel="comb_kernel + prep_kernel/ladder_kernel<ENDO, PREBUILT, DH> + normalize_kernel<8>",
                 text="BASELINE.json configs[3]: 2^22 dh_exchange = DH_endo(a, DH_endo(b, G)) over 8 GPUs, i.e. 2^19 exchanges per GPU "
                      "(first half fixed-base through the 80-point comb of [392]G, same affine outputs as with table_endo([392]G); "
                      "second half variable-base); affine in/out"),
    "cfg5": dict(batch=1 << 17, bytes=(192 + 352) // 2, mads=(83_300 + 97_600) // 2, kernel="partition_kernel + prep_kernel<ENDO> + ladder_kernel<ENDO, PREBUILT> with a per-lane table pointer",
                 text="BASELINE.json configs[4]: mixed batch 2^20 over 8 GPUs, i.e. 2^17 per GPU, 50% fixed-base / 50% variable-base MUL_endo"),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=500)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="cfg2", help="BASELINE.json configuration (default: the headline, cfg2)")
    ap.add_argument("--batch", type=int, default=0, help="units per GPU per step (default: the workload's BASELINE size)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist
    from fourq_amd import Engine, codec, constants
    from fourq_amd.dist import gather_rows, init_process_group

    # FOURQ_BENCH_REHEARSE=1: every rank on GPU 0 over gloo -- exercises the N>1 code path on a one-GPU box
    rehearse = os.environ.get("FOURQ_BENCH_REHEARSE") == "1"
    rank, local_rank, world = init_process_group("gloo" if rehearse else "nccl")
    if rehearse:
        local_rank = 0
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d" % (args.gpus, world, args.gpus))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    stream = torch.cuda.Stream(device=dev)     # a real (non-null) stream: the engine launches on it, the events time it
    torch.cuda.set_stream(stream)
    eng = Engine(local_rank, stream=stream.cuda_stream)

    wl = WORKLOADS[args.workload]
    n = args.batch or wl["batch"]
    to_dev = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int64)).to(dev)
    # ---- synthetic workload, generated on the GPU: rank r uses seeds base+2r, base+1+2r
    seed = {"cfg2": 20002, "cfg3": 30002, "cfg4": 40002, "cfg5": 50002}[args.workload] + 2 * rank
    G_aff = (constants.Gx, constants.Gy)
    g1 = codec.pack_point(G_aff + ((1, 0),) + G_aff)
    table_g = eng.table_endo(g1)
    scalars_h = seeded_scalars(seed, n)
    scalars = to_dev(scalars_h)
    second_h = seeded_scalars(seed + 1, n)
    second = to_dev(second_h)
    points = torch.empty((n, 20), dtype=torch.int64, device=dev)
    eng.mul_endo_fixed_dev(second, table_g, points, n)          # P_i = [k_i]G, raw R1 (projective, Z != 1)
    out = torch.empty((n, 20), dtype=torch.int64, device=dev)
    extra_h = None
    if args.workload == "cfg2":
        def step():
            eng.mul_endo_dev(scalars, points, out, n)
    elif args.workload == "cfg3":
        extra_h = eng.table_windowed(g1)

        def step():
            eng.mul_windowed_fixed_dev(scalars, extra_h, out, n)
    elif args.workload == "cfg4":
        g392 = eng.mul_endo(codec.pack_scalars([392]), g1.reshape(1, 20))[0]      # curve4q.py:758
        extra_h = eng.table_endo(g392)
        g_aff = to_dev(np.repeat(codec.pack_point(G_aff).reshape(1, 8), n, axis=0))
        mid = torch.empty((n, 8), dtype=torch.int64, device=dev)
        out = torch.empty((n, 8), dtype=torch.int64, device=dev)
        st1 = torch.empty(n, dtype=torch.uint8, device=dev)
        st2 = torch.empty(n, dtype=torch.uint8, device=dev)

        comb_h = eng.comb_table(g392)                                     # 80-point comb of [392]G (draft :725-729)

        def step():
            eng.comb_mul_dev(second, comb_h, mid, st1, n)             # == DH_endo(b, G, table_endo([392]G)), affine
            eng.dh_endo_dev(scalars, mid, None, out, st2, n)          # DH_endo(a, .)
    else:
        extra_h = table_g
        flags_h = (np.frombuffer(random.Random(seed + 7).getrandbits(8 * n).to_bytes(n, "little"), dtype=np.uint8) & 1).copy()
        flags = torch.from_numpy(flags_h).to(dev)

        def step():
            eng.mul_endo_mixed_dev(scalars, points, flags, extra_h, out, n)
    torch.cuda.synchronize()

    # The clock governor needs ~35 ms of load to reach the sustained clock (tools/clock_ramp.py,
    # profiles/clock_ramp_r01.txt: 0.43 ms per launch cold, 0.364 ms from launch 100 on).  Throughput is a
    # sustained-rate metric, so the device is brought to that state before the W warm-up steps; untimed.
    settle_ms = float(os.environ.get("FOURQ_BENCH_SETTLE_MS", "80"))
    t_settle = time.perf_counter()
    while (time.perf_counter() - t_settle) * 1e3 < settle_ms:
        for _ in range(8):
            step()
        torch.cuda.synchronize()
    for _ in range(args.warmup):
        step()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ev0.record(stream)                      # HIP events on the launch stream, around the K timed steps
    for _ in range(args.steps):
        step()
    ev1.record(stream)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    kernel_ms = ev0.elapsed_time(ev1) / max(1, args.steps)      # average launch duration, inter-launch gaps included

    t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if rehearse else dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())

    # ---- plumbing check of the only collective the path has: gather the results once (untimed)
    gathered = gather_rows(out, n * world, dst=0) if world > 1 else out
    if rank == 0:
        assert gathered.shape[0] == n * world
        total = n * world * args.steps
        value = total / elapsed
        ach_gbs = wl["bytes"] * n / (kernel_ms * 1e-3) / 1e9
        mad_rate = wl["mads"] * n / (kernel_ms * 1e-3)
        line = {
            "metric": "FourQ scalar-mults/sec (batch, whole node)", "value": round(value, 1), "unit": "scalar-mults/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
            "config": {"workload": wl["text"], "batch_per_gpu": n, "parallelism": "independent shards x%d, no data-path collective" % world,
                       "clock_settle_ms": settle_ms},
            "roofline": {"bound": "hbm", "achieved": round(ach_gbs, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(ach_gbs / HBM_PEAK_GBS, 6), "traffic": _pmc_traffic() if args.workload == "cfg2" else None,
                         "kernel": wl["kernel"], "kernel_ms": round(kernel_ms, 4), "algorithmic_bytes_per_launch": wl["bytes"] * n,
                         "note": "the path is integer-VALU bound, not HBM bound (SURVEY.md 8d): see valu_roofline"},
            "valu_roofline": {"bound": "valu-int (v_mad_u64_u32, 4 cycles per wave64 per SIMD)", "achieved": round(mad_rate / 1e12, 3),
                              "peak": round(VALU_MAD_PEAK / 1e12, 3), "unit": "Tmad/s", "frac": round(mad_rate / VALU_MAD_PEAK, 4)},
        }
        if args.workload == "cfg4":
            line["config"]["note"] = "one unit = one exchange = two DH_core evaluations"
        if world == 1 and not args.no_cpu_baseline:
            ms = codec.unpack_scalars(scalars_h)
            out_h = out.cpu().numpy().view(np.uint64)
            expected = codec.unpack_points(out_h)
            if args.workload == "cfg2":
                pts = codec.unpack_points(points.cpu().numpy().view(np.uint64))
                line["cpu_baseline"] = cpu_baseline("mul_endo", list(zip(ms, pts)), None, expected, "MUL_endo(m, P)")
            elif args.workload == "cfg3":
                line["cpu_baseline"] = cpu_baseline("mul_windowed_fixed", ms, codec.unpack_table(extra_h), expected, "MUL_windowed(m, G, table)")
            elif args.workload == "cfg4":
                if bool(st1.any()) or bool(st2.any()):
                    raise SystemExit("cfg4: unexpected DH failure status")
                bs = codec.unpack_scalars(second_h)
                line["cpu_baseline"] = cpu_baseline("dh_exchange", list(zip(ms, bs)), codec.unpack_table(extra_h), expected,
                                                    "DH_endo(a, DH_endo(b, G, table))")
            else:
                pts = codec.unpack_points(points.cpu().numpy().view(np.uint64))
                items = [(m, P if f else None) for m, P, f in zip(ms, pts, flags_h)]
                line["cpu_baseline"] = cpu_baseline("mixed", items, codec.unpack_table(extra_h), expected, "50/50 fixed/variable MUL_endo")
            line["cpu_baseline"]["c_restatement"] = c_port_baseline(
                args.workload, scalars_h, second_h, points.cpu().numpy().view(np.uint64), extra_h, flags_h if args.workload == "cfg5" else None, out_h)
            line["cpu_baseline"]["sample"] += "; plus %d edge-case scalar/point pairs (0, 1, 2, N-1, N, N+1, 2N, 2^255, 2^256-1 on +-G), exact" % edge_case_check(eng)
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
