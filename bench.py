#!/usr/bin/env python3
"""bench.py -- FourQ scalar-mults/sec: BASELINE.json config 2 per GPU as the headline, configs 3-5 beside it.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload cfg2|cfg3|cfg4|cfg5]

`--gpus N` with N > 1 started from a bare shell launches its own N rank processes (python -m
torch.distributed.run on 127.0.0.1, a free port) BEFORE anything touches a GPU and relays rank 0's line; started
under torch.distributed.run (RANK in the environment) it is one of the ranks.  FOURQ_BENCH_REHEARSE=1 puts every
rank on GPU 0 over gloo (rehearsal of the N > 1 code on a one-GPU box; at most 6 ranks).

One step = one pass of the hot path over one batch: 2^16 variable-base MUL_endo(m_i, P_i) per GPU (random 256-bit
scalars; P_i = projective N-torsion points, raw R1 outputs of fixed-base [k_i]G), inputs and outputs resident in
HBM.  The default run then measures configs 3, 4, 5 at their per-GPU sizes (nested under "configs") and, on one GPU,
the PCIe-inclusive rate of the host-array API ("pcie_inclusive") and the CPU baseline.  Ranks are independent (weak
scaling, no data-path collective); the one collective of the path, the gather of results, is timed apart
("gather_ms").  Every rank checks every output of its shard against the C restatement of the reference before
anything is reported.  Rank 0 prints ONE JSON line.  Synthetic data; integer arithmetic (u32 x 5 limbs of
GF(2^127-1) inside, "u64" words at the ABI).
"""
import argparse
import json
import os
import random
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0                # MI355X_MICROARCH.md
VALU_MAD_PEAK = 1024 * 64 / 4.0 * 2.4e9    # nominal: a wave64 v_mad_u64_u32 occupies a SIMD for 4 cycles, 1024 SIMDs, 2.4 GHz peak clock
# What the silicon sustains on a stream of independent v_mad_u64_u32 with every SIMD busy (tools/microbench/true_rates.hip,
# profiles/true_rates_r01.txt): 1.84 ns per wave64 instruction per SIMD at 8 waves (1.99 at 4) -- the chip does not hold
# 2.4 GHz under this load.  Fractions against this figure say how far a kernel is from what the multiplier can really deliver.
VALU_MAD_PEAK_MEASURED = 1024 * 64 / 1.84e-9

# Per workload: algorithmic bytes per unit and algorithmic 32x32->64 multiply-adds per unit (both SURVEY.md 8d: GF(p)
# mul = 16, GF(p^2) M = 48, S = 32 on the reference's real M/S counts), and the multiply-adds this implementation
# EXECUTES per unit in radix 2^26 (M = 100, S = 50; DBL = 3M+4S = 500, ADD = 8M = 800; ladder step 1 300 x 64 endo,
# 4 DBL + ADD = 2 800 x 62 windowed; table_endo 14 300; DH extras 8 600 (6 800 with the shared inversion); comb
# 6 DBL + 27 mixed ADD (7M) = 21 900 + its share of an inversion): DESIGN.md section 5.
# `text` (<= 120 characters) and `kernel_name` go into the line; what they abbreviate is spelt out in profiles/BENCH_FIELDS.md.
WORKLOADS = {
    "cfg2": dict(batch=1 << 16, steps=500, bytes=32 + 160 + 160, alg_mads=49_440, mads=97_600, seed=20002,
                 kernel_name="ladder_kernel<ENDO,FUSED>", unit="MUL_endo(m, P), variable base",
                 text="BASELINE.json configs[1]: 2^16 variable-base MUL_endo(m,P) per GPU, random 256-bit scalars, raw R1 in/out in HBM"),
    "cfg3": dict(batch=1 << 20, steps=60, bytes=32 + 160, alg_mads=91_264, mads=173_600, seed=30002,
                 kernel_name="ladder_kernel<WINDOWED,LDS>", unit="MUL_windowed(m, G, table)",
                 text="BASELINE.json configs[2]: 2^20 fixed-base MUL_windowed(m,G,table) per GPU, table in LDS, raw R1 out"),
    # cfg4: alg_mads prices BOTH halves at the reference's algorithm (DH_endo with table_endo([392]G): 47 616; DH_endo variable
    # base: 55 072).  The keygen half actually runs the comb (6 DBL + 27 mixed ADD of 7 M: 6 x 272 + 27 x 336 = 10 704 units
    # + its share of an inversion ~ 400), so alg_mads_run = 11 104 + 55 072 is the figure comparable with the executed one.
    "cfg4": dict(batch=1 << 19, steps=60, bytes=2 * 161, alg_mads=47_616 + 55_072, alg_mads_run=11_104 + 55_072, mads=22_500 + 97_600 + 6_800, seed=40002,
                 kernel_name="comb_kernel + ladder_kernel<ENDO,FUSED,DH> + 2 x normalize_kernel<8>",
                 unit="exchange = DH_endo(a, DH_endo(b, G)): two DH_core evaluations",
                 text="BASELINE.json configs[3]: 2^22 dh_exchange over 8 GPUs = 2^19 DH_endo(a, DH_endo(b,G)) per GPU, affine in/out"),
    "cfg5": dict(batch=1 << 17, steps=300, bytes=(192 + 352) // 2, alg_mads=(41_984 + 49_440) // 2, mads=(83_300 + 97_600) // 2, seed=50002,
                 kernel_name="partition_kernel + mixed_queue_kernel (persistent, device-side work queue)",
                 unit="MUL_endo, 50% fixed base / 50% variable base",
                 text="BASELINE.json configs[4]: mixed 2^20 over 8 GPUs = 2^17 per GPU, 50% fixed / 50% variable MUL_endo, persistent kernel"),
}
assert all(len(w["text"]) <= 120 for w in WORKLOADS.values())
RANK_SEED_STRIDE = 16        # rank r draws from seed + 16 r (rank 0 = the seeds of SURVEY.md 8d)


def seeded_scalars(seed, n):
    import numpy as np
    rng = random.Random(seed)
    return np.frombuffer(rng.getrandbits(256 * n).to_bytes(32 * n, "little"), dtype="<u8").reshape(n, 4).copy()


def seeded_flags(seed, n):
    import numpy as np
    return (np.frombuffer(random.Random(seed).getrandbits(8 * n).to_bytes(n, "little"), dtype=np.uint8) & 1).copy()


CORES_CAP = int(os.environ.get("FOURQ_BENCH_CORES", "16"))      # one rank's share of the host on the GPU boxes: 16 cores per GPU
PER_CORE_REFERENCE_SURVEY = 410.0    # MUL_endo/s per core of the REFERENCE's own Python 2 code under python3 in the build container


def host_cores_granted():
    """Cores this process may really use: scheduler affinity capped by the cgroup CPU quota (the GPU box shows every core of the host
    but grants a share of them)."""
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
    return cores


def host_cores(cap=True):
    """What the CPU legs USE: the granted cores, capped at CORES_CAP (one rank's share).  cap=False: the grant itself (the ranks of an
    N-GPU job divide what the whole job was granted)."""
    return min(host_cores_granted(), CORES_CAP) if cap else host_cores_granted()


# ---- CPU baseline (the oracle is the checker and the thing timed here, never part of the GPU path) --------------
def _cpu_worker(args):
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import curve4q_oracle as o
    kind, items, extra = args
    G = (o.Gx, o.Gy)
    G1 = o.AffineToR1(o.Gx, o.Gy)       # with a table the reference ignores the point (curve4q.py:209, :426)
    t0 = time.perf_counter()
    if kind == "cfg2":
        outs = [o.MUL_endo(m, P) for m, P in items]
    elif kind == "cfg3":
        outs = [o.MUL_windowed(m, G1, table=extra) for m in items]
    elif kind == "cfg4":
        outs = [o.DH_endo(a, o.DH_endo(b, G, table=extra)) for a, b in items]
    else:  # cfg5: (m, P or None)
        outs = [o.MUL_endo(m, P) if P is not None else o.MUL_endo(m, G1, table=extra) for m, P in items]
    return time.perf_counter() - t0, outs


def cpu_items(workload, d, sample):
    """(items, extra, text) of the Python-oracle leg: the first `sample` units of one workload's batch."""
    from fourq_amd import codec
    ms = codec.unpack_scalars(d["scalars_h"][:sample])
    if workload == "cfg2":
        return list(zip(ms, codec.unpack_points(d["points_h"][:sample]))), None, "MUL_endo(m, P)"
    if workload == "cfg3":
        return ms, codec.unpack_table(d["table_h"]), "MUL_windowed(m, G, table)"
    if workload == "cfg4":
        import curve4q_oracle as o
        t392 = o.table_endo(o.MUL_endo(392, o.AffineToR1(o.Gx, o.Gy)))
        return list(zip(ms, codec.unpack_scalars(d["second_h"][:sample]))), t392, "DH_endo(a, DH_endo(b, G, table))"
    pts = codec.unpack_points(d["points_h"][:sample])
    return [(m, P if f else None) for m, P, f in zip(ms, pts, d["flags_h"][:sample])], codec.unpack_table(d["table_h"]), "50/50 fixed/variable MUL_endo"


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
    # cores = the processes actually used; the box's total, the share this process was granted and the cap are stated beside it
    return {"value": round(total / busy, 1), "unit": "scalar-mults/s", "cores": cores, "host_cores_total": os.cpu_count(), "host_cores_granted": host_cores_granted(),
            "cores_cap": CORES_CAP, "kind": "port",
            "sample": "%d units of the timed batch (%s), oracle/curve4q_oracle.py, %d per core x %d processes, %.1f s wall, outputs == GPU's" % (total, what, per_core, cores, wall),
            "per_core": round(per_core / busy, 1),
            # the port is a baseline, never the target -- and it is faster per core than the code it restates (python3 big ints, no op counters)
            "per_core_reference_survey": PER_CORE_REFERENCE_SURVEY}


def c_oracle_gate(workload, data, got_words, world=1):
    """The C restatement (oracle/fourq_oracle.c, OpenMP) over the WHOLE batch of this rank: the parity gate on every
    output and, for context, its rate on the host cores.  Every rank of an N-GPU job runs its own gate at
    the same time, so each takes its share of the host's cores.  Returns (units per second, threads, expected words)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import numpy as np
    import oracle_c as oc
    from fourq_amd import codec, constants
    share = host_cores() if world <= 1 else max(1, min(host_cores(), host_cores(cap=False) // world))
    threads = oc.set_num_threads(share)
    assert threads == share, "OpenMP did not take the thread count"
    s, k = data["scalars_h"], data["second_h"]
    t0 = time.perf_counter()
    if workload == "cfg2":
        want = oc.mul(oc.ENDO, s, data["points_h"])
    elif workload == "cfg3":
        want = oc.mul(oc.WINDOWED, s, None, data["table_h"])
    elif workload == "cfg4":
        g = np.repeat(codec.pack_point((constants.Gx, constants.Gy)).reshape(1, 8), len(s), axis=0)
        mid, st1 = oc.dh(oc.ENDO, k, g)
        want, st2 = oc.dh(oc.ENDO, s, mid)
        if st1.any() or st2.any():
            raise SystemExit("cfg4: unexpected DH failure status in the C oracle")
    else:
        fixed = oc.mul(oc.ENDO, s, None, data["table_h"])
        var = oc.mul(oc.ENDO, s, data["points_h"])
        want = np.where(data["flags_h"].reshape(-1, 1) != 0, var, fixed)
    dt = time.perf_counter() - t0
    if not np.array_equal(want, got_words):
        raise SystemExit("PARITY FAILURE: GPU result differs from the C oracle on the full batch (%s)" % workload)
    units = len(s) * (2 if workload in ("cfg4", "cfg5") else 1)     # cfg5 evaluates both variants, cfg4 both halves
    return units / dt, threads, want


def edge_case_check(eng):
    """every run carries an edge-case mini-batch -- scalars {0,1,2,N-1,N,N+1,2N,2^255,2^256-1}
    on G and -G through MUL_endo, MUL_windowed and DH_endo, bit-exact against the Python oracle (the oracle is the
    checker; part of every rank's parity gate).  Returns the number of cases."""
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


def valu_roofline(wl, n, kernel_ms):
    """Multiply-adds per second against the multiplier's ceiling: nominal (4 cycles per wave64 v_mad_u64_u32 at 2.4 GHz) and
    measured (what a saturating stream of them really sustains on this silicon)."""
    per_s = n / (kernel_ms * 1e-3)
    rec = {"bound": "valu-int (v_mad_u64_u32, 4 cycles per wave64 per SIMD)", "peak": round(VALU_MAD_PEAK / 1e12, 3),
           "peak_measured": round(VALU_MAD_PEAK_MEASURED / 1e12, 3), "unit": "Tmad/s",
           "algorithmic_mads_per_unit": wl["alg_mads"], "executed_mads_per_unit": wl["mads"],
           "algorithmic_frac": round(wl["alg_mads"] * per_s / VALU_MAD_PEAK, 4), "executed_frac": round(wl["mads"] * per_s / VALU_MAD_PEAK, 4),
           "algorithmic_frac_of_measured_peak": round(wl["alg_mads"] * per_s / VALU_MAD_PEAK_MEASURED, 4),
           "executed_frac_of_measured_peak": round(wl["mads"] * per_s / VALU_MAD_PEAK_MEASURED, 4)}
    if "alg_mads_run" in wl:                 # cfg4: the keygen half runs the comb, not the reference's 64-step ladder (profiles/BENCH_FIELDS.md)
        rec["algorithmic_mads_per_unit_of_the_algorithm_run"] = wl["alg_mads_run"]
        rec["algorithmic_frac_of_the_algorithm_run"] = round(wl["alg_mads_run"] * per_s / VALU_MAD_PEAK, 4)
    return rec


# ---- one workload on this rank ---------------------------------------------------------------------------------
class _Cuda:
    """The few things the rank body needs from torch.cuda and from the engine, behind one object: tests/test_dist.py replaces it
    with a CPU stand-in to run the N > 1 body (barriers, the MAX over ranks, the gather, the all-ranks parity reduction, the JSON
    line) over gloo where there is no GPU.  Product runs never see anything but this class."""

    def set_device(self, index):
        import torch
        torch.cuda.set_device(index)

    def device(self, index):
        import torch
        return torch.device("cuda", index)

    def new_stream(self, device):
        import torch
        stream = torch.cuda.Stream(device=device)      # a real (non-null) stream: the engine launches on it, the events time it
        torch.cuda.set_stream(stream)
        return stream

    def synchronize(self):
        import torch
        torch.cuda.synchronize()

    def event(self):
        import torch
        return torch.cuda.Event(enable_timing=True)

    def empty_cache(self):
        import torch
        torch.cuda.empty_cache()

    def engine(self, index, stream):
        from fourq_amd import Engine
        return Engine(index, stream=stream.cuda_stream)


GPU = _Cuda()


class Bench:
    def __init__(self, rank, local_rank, world, rehearse, dist_on=None):
        from fourq_amd import codec, constants
        self.rank, self.world, self.rehearse = rank, world, rehearse
        # the rank body's collectives run whenever a process group exists: N > 1, or FOURQ_BENCH_FORCE_DIST=1 (a group of ONE rank over nccl:
        # the only way to drive the RCCL branch of this file on a one-GPU box; tests/test_gpu_multi.py)
        self.dist_on = (world > 1) if dist_on is None else dist_on
        GPU.set_device(local_rank)
        self.local_rank = local_rank
        self.dev = GPU.device(local_rank)
        self.stream = GPU.new_stream(self.dev)
        self.eng = GPU.engine(local_rank, self.stream)
        G_aff = (constants.Gx, constants.Gy)
        self.g1 = codec.pack_point(G_aff + ((1, 0),) + G_aff)
        self.g_aff = codec.pack_point(G_aff)
        self.table_g = self.eng.table_endo(self.g1)
        self.settle_ms = float(os.environ.get("FOURQ_BENCH_SETTLE_MS", "80"))
        self.clock_mode = os.environ.get("FOURQ_BENCH_CLOCK", "bracket")
        if self.clock_mode == "bracket" and hasattr(self.eng, "diag_clock_begin"):
            self.eng.diag_clock_begin()              # sizes the probe's buffers outside any timed region
            self.eng.diag_clock_end()

    def to_dev(self, a):
        import numpy as np
        import torch
        a = np.ascontiguousarray(a)
        return torch.from_numpy(a.view(np.int64) if a.dtype == np.uint64 else a).to(self.dev)

    def prepare(self, workload, n):
        """Synthetic inputs of SURVEY.md 8d, generated on the GPU; returns (step function, data dict)."""
        import numpy as np
        import torch
        eng, wl = self.eng, WORKLOADS[workload]
        seed = wl["seed"] + RANK_SEED_STRIDE * self.rank
        d = {}
        if workload == "cfg5":                       # flags: seed 50002, points: 50003, scalars: 50004
            d["flags_h"] = seeded_flags(seed, n)
            d["second_h"] = seeded_scalars(seed + 1, n)
            d["scalars_h"] = seeded_scalars(seed + 2, n)
        else:
            d["scalars_h"] = seeded_scalars(seed, n)
            d["second_h"] = seeded_scalars(seed + 1, n)
        scalars, second = self.to_dev(d["scalars_h"]), self.to_dev(d["second_h"])
        out = torch.empty((n, 20), dtype=torch.int64, device=self.dev)
        if workload in ("cfg2", "cfg5"):
            points = torch.empty((n, 20), dtype=torch.int64, device=self.dev)
            eng.mul_endo_fixed_dev(second, self.table_g, points, n)          # P_i = [k_i]G, raw R1 (projective, Z != 1)
            GPU.synchronize()
            d["points_h"] = points.cpu().numpy().view(np.uint64)
        if workload == "cfg2":
            def step():
                eng.mul_endo_dev(scalars, points, out, n)
        elif workload == "cfg3":
            d["table_h"] = eng.table_windowed(self.g1)

            def step():
                eng.mul_windowed_fixed_dev(scalars, d["table_h"], out, n)
        elif workload == "cfg4":
            from fourq_amd import codec
            g392 = eng.mul_endo(codec.pack_scalars([392]), self.g1.reshape(1, 20))[0]      # curve4q.py:758
            comb_h = eng.comb_table(g392)                                     # 1024-point comb of [392]G (draft :725-729)
            eng.comb_stage(comb_h)                                            # uploaded once; the steps pass None = "the staged table"
            eng.reserve(n)                                                    # the steps only enqueue (fourq_ctx_reserve)
            mid = torch.empty((n, 8), dtype=torch.int64, device=self.dev)
            out = torch.empty((n, 8), dtype=torch.int64, device=self.dev)
            st1 = torch.empty(n, dtype=torch.uint8, device=self.dev)
            st2 = torch.empty(n, dtype=torch.uint8, device=self.dev)
            d["status"] = (st1, st2)

            def step():
                eng.comb_mul_dev(second, None, mid, st1, n)               # == DH_endo(b, G, table_endo([392]G)), affine
                eng.dh_endo_dev(scalars, mid, None, out, st2, n)          # DH_endo(a, .)
        else:
            d["table_h"] = self.table_g
            flags = self.to_dev(d["flags_h"])

            def step():
                eng.mul_endo_mixed_dev(scalars, points, flags, d["table_h"], out, n)
        d["out"] = out
        d["keep"] = (scalars, second)
        return step, d

    def run(self, workload, n, steps, warmup):
        """W warm-up steps, then K timed steps between barrier + synchronize on both sides; MAX over ranks."""
        import torch
        import torch.distributed as dist
        wl = WORKLOADS[workload]
        step, d = self.prepare(workload, n)
        GPU.synchronize()
        if self.dist_on:
            # Align the ranks BEFORE the clock is settled: a rank that reaches the contract's barrier (below) milliseconds ahead of the slowest
            # one idles there while its clock falls back, and the MAX over ranks then times that rank's ramp instead of its kernels.  After
            # this barrier every rank does the same time-based settle loop and the same W warm-up steps, and they arrive together.
            dist.barrier()
        # The clock governor needs ~35 ms of load to reach the sustained clock (tools/clock_ramp.py,
        # profiles/clock_ramp_r01.txt: 0.43 ms per launch cold, 0.364 ms from launch 100 on).  Throughput is a
        # sustained-rate metric, so the device is brought to that state before the W warm-up steps; untimed.
        t_settle = time.perf_counter()
        while (time.perf_counter() - t_settle) * 1e3 < self.settle_ms:
            for _ in range(8 if n <= 1 << 17 else 1):
                step()
            GPU.synchronize()
        for _ in range(warmup):
            step()
        ev0, ev1 = GPU.event(), GPU.event()
        GPU.synchronize()
        if self.dist_on:
            dist.barrier()
        GPU.synchronize()
        # The shader clock OF the timed steps: two launches of a stamp kernel on the launch stream, one before the first
        # timed step and one behind the last, whose waves record their CU's cycle counter and the 100 MHz counter (fourq_diag_clock_begin /
        # _stop; paired per CU by _end).  Both lie OUTSIDE the two events; nothing stays resident beside the steps and the product kernels
        # carry no stamp.  FOURQ_BENCH_CLOCK=after|off: the round-5 probe on a backlog behind the timed region / none.
        bracket = self.clock_mode == "bracket" and hasattr(self.eng, "diag_clock_begin")
        t0 = time.perf_counter()
        if bracket:
            self.eng.diag_clock_begin()
        ev0.record(self.stream)                      # HIP events on the launch stream, around the K timed steps
        for _ in range(steps):
            step()
        ev1.record(self.stream)
        if bracket:
            self.eng.diag_clock_stop()               # enqueues; the host does not wait here
        GPU.synchronize()
        if self.dist_on:
            dist.barrier()
        GPU.synchronize()
        elapsed = time.perf_counter() - t0
        kernel_ms = ev0.elapsed_time(ev1) / max(1, steps)      # average launch duration, inter-launch gaps included
        clock = None
        c = None
        if bracket:
            try:
                c = self.eng.diag_clock_end()
            except Exception as exc:                 # a diagnostic must never cost the run its line: no clock, no cycles_per_unit
                print("bench.py: clock bracket unavailable (%s)" % exc, file=sys.stderr)
        if c is not None:
            # the window must be the timed region and nothing else: within 2 % + 0.2 ms of the events' span
            span_ms = kernel_ms * steps
            covered = c["window_us"] * 1e-3 / span_ms if span_ms > 0 else 0.0
            clock = {"in_kernel_mhz": round(c["mhz"], 1), "min_mhz": round(c["mhz_min"], 1), "max_mhz": round(c["mhz_max"], 1),
                     "window_ms": round(c["window_us"] * 1e-3, 3), "mode": "bracket", "window_over_timed_span": round(covered, 4),
                     "valid": bool(abs(c["window_us"] * 1e-3 - span_ms) <= 0.02 * span_ms + 0.2)}
        if self.dist_on:
            t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if self.rehearse else self.dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        if workload == "cfg4" and (bool(d["status"][0].any()) or bool(d["status"][1].any())):
            raise SystemExit("cfg4: unexpected DH failure status")
        if clock is None and self.clock_mode != "off":
            clock = self.clock_under_load(step, kernel_ms)       # a diagnostic launch AFTER the timed region: nothing of it is in `elapsed`
        usable = bool(clock) and clock.get("valid", True)
        total = n * self.world * steps
        ach_gbs = wl["bytes"] * n / (kernel_ms * 1e-3) / 1e9
        traffic, traffic_src = _pmc_traffic(workload, self.eng.build_id)
        rec = {
            "workload": wl["text"], "unit_of_work": wl["unit"], "batch_per_gpu": n, "steps": steps, "warmup": warmup,
            "value": round(total / elapsed, 1), "unit": "scalar-mults/s" if workload != "cfg4" else "exchanges/s",
            "ms_per_step": round(1e3 * elapsed / steps, 4),
            "roofline": {"bound": "hbm", "achieved": round(ach_gbs, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(ach_gbs / HBM_PEAK_GBS, 6),
                         "traffic": None if self.eng.ct_select else traffic, "traffic_source": traffic_src,     # the PMC passes ran on the default kernels
                         "kernel": wl["kernel_name"], "kernel_ms": round(kernel_ms, 4), "algorithmic_bytes_per_launch": wl["bytes"] * n},
            "valu_roofline": valu_roofline(wl, n, kernel_ms),
            # What makes this line comparable across boxes: MI355X devices hold clocks several percent apart under the
            # same kernel, so the same code gives 2.14 - 2.24 x 10^8/s on different boxes; in cycles they agree.
            "clock": clock,
            "cycles_per_unit": None if not usable else round(kernel_ms * 1e-3 * clock["in_kernel_mhz"] * 1e6 / n, 3),
            "kernel_cycles_per_step": None if not usable else round(kernel_ms * 1e-3 * clock["in_kernel_mhz"] * 1e6, 0),
        }
        # The bound that actually holds (round 4, profiles/r04_ladder_step.txt): with every 8-byte instruction 8-byte aligned a SIMD issues
        # ONE wave64 VALU instruction per ~4 cycles for this instruction mix, multiply-add or not, at one, two or four waves per SIMD.
        wave_instr = None if self.eng.ct_select else _pmc_valu_instructions(workload, self.eng.build_id)
        issue_peak = 1024 * 2.4e9 / 4.0                      # wave64 VALU instructions per second, chip-wide, nominal clock
        rec["valu_roofline"]["issue"] = {
            "bound": "valu-issue: one wave64 VALU instruction per SIMD per 4 cycles", "peak": round(issue_peak / 1e9, 1), "unit": "G wave-instructions/s",
            "wave_instructions_per_step": wave_instr, "instructions_per_unit": None if wave_instr is None else round(wave_instr * 64 / n, 1),
            "achieved": None if wave_instr is None else round(wave_instr / (kernel_ms * 1e-3) / 1e9, 1),
            "frac": None if wave_instr is None else round(wave_instr / (kernel_ms * 1e-3) / issue_peak, 4),
            "source": "profiles/pmc_traffic.json"}
        return rec, d

    def clock_under_load(self, step, kernel_ms, window_ms=25.0):
        """Round 5's form of the clock probe (FOURQ_BENCH_CLOCK=after): the stream gets a backlog of steps three windows long (the _dev calls
        only enqueue), and while it drains the probe times `window_ms` on a side stream (fourq_diag_clock).  `valid` is False when the
        stream had already drained when the window closed -- the reading is then an idle chip's."""
        if not hasattr(self.eng, "diag_clock"):
            return None
        backlog = max(4, int(3.0 * window_ms / max(kernel_ms, 1e-3)) + 1)
        for _ in range(backlog):
            step()
        c = self.eng.diag_clock(int(window_ms * 1000))
        GPU.synchronize()
        return {"in_kernel_mhz": round(c["mhz"], 1), "min_mhz": round(c["mhz_min"], 1), "max_mhz": round(c["mhz_max"], 1), "window_ms": window_ms,
                "mode": "after", "steps_queued_behind_the_probe": backlog, "valid": bool(c.get("under_load", True))}

    def parity_gate(self, workload, d):
        """Every output of this rank's shard against the C oracle; raises on any difference."""
        import numpy as np
        got = d["out"].cpu().numpy().view(np.uint64)
        rate, threads, want = c_oracle_gate(workload, d, got, self.world)
        return {"gate": "every output of the shard bit-exact vs oracle/fourq_oracle.c", "ok": True, "units": int(len(got)),
                "c_oracle_units_per_s": round(rate, 1), "c_oracle_threads": threads}, got, want

    def alongside(self, workload, d, steps):
        """cfg2 reports MUL_windowed on the same inputs alongside MUL_endo, cfg3 reports MUL_endo(m, G, table)
        alongside MUL_windowed.  Same timing discipline (HIP events on the launch stream around back-to-back steps), every
        output compared with the C oracle; multiply-add counts as in WORKLOADS."""
        import numpy as np
        import torch
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import oracle_c as oc
        eng, n = self.eng, len(d["scalars_h"])
        scalars = d["keep"][0]
        out = torch.empty((n, 20), dtype=torch.int64, device=self.dev)
        if workload == "cfg2":
            points = self.to_dev(d["points_h"])
            name, alg, mads = "MUL_windowed(m, P), variable base, same scalars and points", 94_992, 181_000
            step = lambda: eng.mul_windowed_dev(scalars, points, out, n)
            want = lambda: oc.mul(oc.WINDOWED, d["scalars_h"], d["points_h"])
        else:
            name, alg, mads = "MUL_endo(m, G, table), fixed base, same scalars", 41_984, 83_300
            step = lambda: eng.mul_endo_fixed_dev(scalars, self.table_g, out, n)
            want = lambda: oc.mul(oc.ENDO, d["scalars_h"], None, self.table_g)
        for _ in range(max(2, steps // 10)):
            step()
        ev0, ev1 = GPU.event(), GPU.event()
        GPU.synchronize()
        ev0.record(self.stream)
        for _ in range(steps):
            step()
        ev1.record(self.stream)
        GPU.synchronize()
        ms = ev0.elapsed_time(ev1) / steps
        if not np.array_equal(out.cpu().numpy().view(np.uint64), want()):
            raise SystemExit("PARITY FAILURE: alongside op of %s differs from the C oracle" % workload)
        return {"op": name, "batch_per_gpu": n, "steps": steps, "ms_per_step": round(ms, 4), "value_per_gpu": round(n / (ms * 1e-3), 1), "unit": "scalar-mults/s",
                "parity_ok": True, "algorithmic_frac": round(alg * n / (ms * 1e-3) / VALU_MAD_PEAK, 4), "executed_frac": round(mads * n / (ms * 1e-3) / VALU_MAD_PEAK, 4)}

    def size_sweep(self, d, sizes=(1, 1024, 16384, 32768, 65536, 65792, 98304), reps=30):
        """How the headline operation's time depends on the batch size (device-resident, HIP events on the launch stream, median of
        `reps` back-to-back calls at a sustained clock).  One lane owns one scalar multiplication for a whole ladder, so the chip works
        in generations of `lanes` elements; batches of at most half a generation and remainders past a generation run two lanes per
        element, those of at most a quarter generation four (pair.hip.h).  The ratio t(65 792) / t(65 536) is VERDICT r2's measure of
        the generation cliff."""
        import torch
        eng = self.eng
        scalars = d["keep"][0]
        points = self.to_dev(d["points_h"])
        n_all = len(d["scalars_h"])
        big = max(sizes)
        if big > n_all:                                   # the sweep's largest size: repeat the batch's inputs
            reps_in = (big + n_all - 1) // n_all
            scalars, points = scalars.repeat(reps_in, 1)[:big].contiguous(), points.repeat(reps_in, 1)[:big].contiguous()
        out = torch.empty((big, 20), dtype=torch.int64, device=self.dev)
        rec = {}
        for n in sizes:
            for _ in range(5):
                eng.mul_endo_dev(scalars, points, out, n)
            times = []
            for _ in range(reps):
                a, b = GPU.event(), GPU.event()
                a.record(self.stream)
                eng.mul_endo_dev(scalars, points, out, n)
                b.record(self.stream)
                GPU.synchronize()
                times.append(a.elapsed_time(b))
            rec[str(n)] = round(sorted(times)[len(times) // 2], 4)
        rec["unit"] = "ms per call of MUL_endo(m, P), variable base, n elements"
        rec["lanes"] = eng.lanes
        if "65536" in rec and "65792" in rec:
            rec["t(65792)/t(65536)"] = round(rec["65792"] / rec["65536"], 3)
        return rec

    def small_batches(self, d, sizes=(1, 1024), reps=30):
        """The other operations of the path at small batch sizes (device-resident, HIP events, median of `reps` calls): what a caller
        pays for ONE reference-shaped call or a small batch -- two / four lanes per element (pair.hip.h), key generation through the comb
        gathered from memory (comb_quad_kernel), a mixed batch with the table chosen per element.  Outputs of the largest size are
        compared with the C oracle once."""
        import numpy as np
        import torch
        from fourq_amd import codec
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import oracle_c as oc
        eng = self.eng
        big = max(sizes)
        scalars, points = d["keep"][0][:big].contiguous(), self.to_dev(d["points_h"][:big])
        s_h, p_h = d["scalars_h"][:big], d["points_h"][:big]
        g_h = np.repeat(self.g_aff.reshape(1, 8), big, axis=0)
        g = self.to_dev(g_h)
        flags_h = (np.arange(big) & 1).astype(np.uint8)
        flags = torch.from_numpy(flags_h).to(self.dev)
        g392 = eng.mul_endo(codec.pack_scalars([392]), self.g1.reshape(1, 20))[0]
        comb_h = eng.comb_table(g392)
        eng.comb_stage(comb_h)
        out = torch.empty((big, 20), dtype=torch.int64, device=self.dev)
        aff = torch.empty((big, 8), dtype=torch.int64, device=self.dev)
        st = torch.empty(big, dtype=torch.uint8, device=self.dev)
        ops = {
            "MUL_windowed(m, P)": lambda n: eng.mul_windowed_dev(scalars, points, out, n),
            "DH_endo(m, Q)": lambda n: eng.dh_endo_dev(scalars, g, None, aff, st, n),
            "keygen: comb of [392]G == DH_endo(m, G)": lambda n: eng.comb_mul_dev(scalars, None, aff, st, n),
            "MUL_endo(m, G, table) fixed base": lambda n: eng.mul_endo_fixed_dev(scalars, self.table_g, out, n),
            "MUL_endo mixed 50/50": lambda n: eng.mul_endo_mixed_dev(scalars, points, flags, self.table_g, out, n),
        }
        rec = {}
        for name, fn in ops.items():
            row = {}
            for n in sizes:
                for _ in range(5):
                    fn(n)
                times = []
                for _ in range(reps):
                    a, b = GPU.event(), GPU.event()
                    a.record(self.stream)
                    fn(n)
                    b.record(self.stream)
                    GPU.synchronize()
                    times.append(a.elapsed_time(b))
                row[str(n)] = round(sorted(times)[len(times) // 2], 4)
            rec[name] = row
        # parity of what was just timed, on `big` elements
        ops["MUL_endo mixed 50/50"](big)
        GPU.synchronize()
        got = out.cpu().numpy().view(np.uint64)
        want = np.where(flags_h[:, None] != 0, oc.mul(oc.ENDO, s_h, p_h), oc.mul(oc.ENDO, s_h, None, self.table_g))
        ok = bool(np.array_equal(got, want))
        ops["keygen: comb of [392]G == DH_endo(m, G)"](big)
        GPU.synchronize()
        kg, kst = aff.cpu().numpy().view(np.uint64), st.cpu().numpy()
        want_kg, want_st = oc.dh(oc.ENDO, s_h, g_h)
        ok = ok and bool(np.array_equal(kg, want_kg)) and bool(np.array_equal(kst, want_st))
        ops["DH_endo(m, Q)"](big)
        GPU.synchronize()
        ok = ok and bool(np.array_equal(aff.cpu().numpy().view(np.uint64), want_kg))
        rec["parity_ok"] = ok
        rec["unit"] = "ms per call of n elements (device-resident)"
        if not ok:
            raise SystemExit("small_batches: outputs differ from the C oracle")
        return rec

    def gather_ms(self, out, n, reps=3):
        """The path's only collective: results gathered to rank 0 (RCCL over xGMI; gloo in rehearsals).  Median of `reps`."""
        import torch
        import torch.distributed as dist
        from fourq_amd.dist import gather_rows
        times, full = [], None
        for _ in range(reps):
            GPU.synchronize()
            dist.barrier()
            t0 = time.perf_counter()
            full = gather_rows(out, n * self.world, dst=0)
            GPU.synchronize()
            times.append((time.perf_counter() - t0) * 1e3)
        if self.rank == 0:
            assert full.shape[0] == n * self.world
        return round(sorted(times)[len(times) // 2], 3)

    def timed_calls(self, call, reps):
        """A synchronous host-array call, timed as the device-resident steps are: the device is first brought to its sustained clock
        (calls repeated for `settle_ms`: the arrays were just generated and pinned by the host, the GPU has been idle meanwhile), then
        `reps` calls are timed one by one.  Returns (median seconds, best seconds, the last call's result)."""
        got = call()                                            # sizes the pipeline's buffers
        t_settle = time.perf_counter()
        while (time.perf_counter() - t_settle) * 1e3 < self.settle_ms:
            got = call()
        times = []
        for _ in range(reps):
            t0 = time.perf_counter()
            got = call()
            times.append(time.perf_counter() - t0)
        times.sort()
        return times[len(times) // 2], times[0], got

    def pcie_inclusive(self, workload, d, want_words, reps, ct_ms=None):
        """SURVEY.md 8(d) wall-clock metric: first H2D byte to last D2H byte through the host-array ABI, from
        host-resident inputs in pinned memory (fourq_host_alloc); the pageable-caller rate beside it.  Outputs are
        compared with the C oracle's."""
        import numpy as np
        eng = self.eng
        n = len(d["scalars_h"])
        pins, extra = [], {}

        def pin(a):
            pins.append(eng.host_array(a))
            return pins[-1]

        def pin_empty(shape, dtype=np.uint64):
            pins.append(eng.host_empty(shape, dtype))
            return pins[-1]

        if workload == "cfg2":
            io, bytes_in, bytes_out = "R1 in / R1 out (the reference's own MUL_* signature)", 192, 160
            s_pin, p_pin, o_pin = pin(d["scalars_h"]), pin(d["points_h"]), pin_empty((n, 20))
            calls = {"pinned": lambda: eng.mul_endo(s_pin, p_pin, out=o_pin), "pageable": lambda: eng.mul_endo(d["scalars_h"], d["points_h"])}
        elif workload == "cfg3":
            io, bytes_in, bytes_out = "scalar in / R1 out (the reference's own MUL_* signature, table resident)", 32, 160
            s_pin, o_pin = pin(d["scalars_h"]), pin_empty((n, 20))
            calls = {"pinned": lambda: eng.mul_windowed_fixed(s_pin, d["table_h"], out=o_pin),
                     "pageable": lambda: eng.mul_windowed_fixed(d["scalars_h"], d["table_h"])}
        elif workload == "cfg4":
            # the host-array exchange call: DH_endo(a, DH_endo(b, G, table_endo([392]G))), public keys never leave the device
            io, bytes_in, bytes_out = "two scalars in / affine shared point + status out (fourq_dh_exchange_batch)", 64, 65
            from fourq_amd import codec
            t392 = eng.table_endo(eng.mul_endo(codec.pack_scalars([392]), self.g1.reshape(1, 20))[0])
            a_pin, b_pin, o_pin, st_pin = pin(d["scalars_h"]), pin(d["second_h"]), pin_empty((n, 8)), pin_empty((n,), np.uint8)

            def exchange(a, b, o, st):
                out, status = eng.dh_exchange(a, b, self.g_aff, t392, out=o, status=st)
                if status.any():
                    raise SystemExit("cfg4 host-array path: unexpected DH failure status")
                return out
            calls = {"pinned": lambda: exchange(a_pin, b_pin, o_pin, st_pin), "pageable": lambda: exchange(d["scalars_h"], d["second_h"], None, None)}

            def exchange_comb(a, b, o, st):        # the keygen half through the staged comb: bench's cfg4 step as ONE host-array call
                out, status = eng.dh_exchange_comb(a, b, None, out=o, status=st)
                if status.any():
                    raise SystemExit("cfg4 host-array path (comb): unexpected DH failure status")
                return out
            extra = {"keygen_through_the_comb (fourq_dh_exchange_comb_batch, pinned)": lambda: exchange_comb(a_pin, b_pin, o_pin, st_pin)}
        else:
            io, bytes_in, bytes_out = "scalar + R1 + flag in / R1 out (fourq_mul_endo_mixed_batch)", 193, 160
            s_pin, p_pin, f_pin, o_pin = pin(d["scalars_h"]), pin(d["points_h"]), pin(d["flags_h"]), pin_empty((n, 20))
            calls = {"pinned": lambda: eng.mul_endo_mixed(s_pin, p_pin, f_pin, d["table_h"], out=o_pin),
                     "pageable": lambda: eng.mul_endo_mixed(d["scalars_h"], d["points_h"], d["flags_h"], d["table_h"])}
        rec = {"io": "%s: %d B in + %d B out per unit" % (io, bytes_in, bytes_out), "batch": n, "reps": reps}
        if workload == "cfg4":
            calls.update(extra)
        for label in calls:
            call = calls[label]
            dt, best, got = self.timed_calls(call, reps)
            if not np.array_equal(got, want_words):
                raise SystemExit("PARITY FAILURE: host-array path (%s, %s) differs from the C oracle" % (workload, label))
            eng.host_timing(True)                               # one more call with the copies timed (not a timed rep: the events are not free)
            call()
            eng.host_timing(False)
            st = eng.host_stats()
            r = {"value": round(n / dt, 1), "ms_per_step": round(dt * 1e3, 4), "best_ms": round(best * 1e3, 4), "gbs_h2d": round(st["gbs_h2d"], 2), "gbs_d2h": round(st["gbs_d2h"], 2),
                 "chunks": st["chunks"], "wall_gbs_both_directions": round(n * (bytes_in + bytes_out) / dt / 1e9, 2),
                 "kernel_stream_busy_ms": round(st.get("kernels_ms", 0.0), 4), "kernel_stream_span_ms": round(st.get("kernels_span_ms", 0.0), 4)}
            if label == "pinned":
                rec.update(r)
            elif label == "pageable":
                rec["pageable_caller"] = r
            else:
                rec[label.split(" ")[0]] = dict(r, call=label)
        if workload == "cfg2":
            # the same operation through the I/O formats that do not waste the link (SURVEY.md 8(d) "affine-only I/O variant"): the same
            # points in affine form / encoded, canonical affine / encoded results; expected words from the C oracle (R1toAffine, encode
            # of its R1 rows -- the affine result of a scalar multiplication does not depend on the input's projective representative)
            sys.path.insert(0, os.path.join(ROOT, "oracle"))
            import oracle_c as oc
            aff_in = oc.r1_to_affine(d["points_h"])
            want_aff = oc.r1_to_affine(want_words)
            want_enc = oc.encode(want_aff)
            a_pin, e_pin = pin(aff_in), pin(oc.encode(aff_in))
            oa_pin, oe_pin, st_pin = pin_empty((n, 8)), pin_empty((n, 32), np.uint8), pin_empty((n,), np.uint8)

            def mul_bytes():
                out, status = eng.mul_bytes(s_pin, e_pin, out=oe_pin, status=st_pin)
                if status.any():
                    raise SystemExit("cfg2 bytes flavour: a point did not decode")
                return out
            variants = {"affine": (lambda: eng.mul_affine(s_pin, a_pin, out=oa_pin), want_aff, 96, 64,
                                   "fourq_mul_endo_affine_batch: scalar + affine point in, canonical affine out (parity level L1)"),
                        "bytes": (mul_bytes, want_enc, 64, 33, "fourq_mul_endo_bytes_batch: scalar + 32-byte point in, 32-byte point + status out")}
            for label, (call, want_v, b_in, b_out, what) in variants.items():
                dt, best, got = self.timed_calls(call, reps)
                if not np.array_equal(got, want_v):
                    raise SystemExit("PARITY FAILURE: host-array path (cfg2, %s I/O) differs from the C oracle" % label)
                st = eng.host_stats()
                rec[label] = {"call": what, "io": "%d B in + %d B out per unit" % (b_in, b_out), "value": round(n / dt, 1), "ms_per_step": round(dt * 1e3, 4),
                              "best_ms": round(best * 1e3, 4), "chunks": st["chunks"], "wall_gbs_both_directions": round(n * (b_in + b_out) / dt / 1e9, 2)}
            # One generation is one chunk: copy in, kernels, copy out in series, whatever the format.  Sixteen generations (the same
            # elements sixteen times over, 2^20) is where the pipeline overlaps them and the format decides: R1 I/O is bound by the link.
            reps_big, big = 7, 16 * n
            sb, pb, ab, eb = (pin(np.tile(x, (16, 1))) for x in (d["scalars_h"], d["points_h"], aff_in, oc.encode(aff_in)))
            ob, oab, oeb, stb = pin_empty((big, 20)), pin_empty((big, 8)), pin_empty((big, 32), np.uint8), pin_empty((big,), np.uint8)
            big_calls = {"r1": (lambda: eng.mul_endo(sb, pb, out=ob), want_words, 352),
                         "affine": (lambda: eng.mul_affine(sb, ab, out=oab), want_aff, 160),
                         "bytes": (lambda: eng.mul_bytes(sb, eb, out=oeb, status=stb)[0], want_enc, 97)}
            rec["at_2^20"] = {"batch": big, "reps": reps_big}
            for label, (call, want_v, nbytes) in big_calls.items():
                dt, best, got = self.timed_calls(call, reps_big)
                if not all(np.array_equal(got[k * n:(k + 1) * n], want_v) for k in range(16)):
                    raise SystemExit("PARITY FAILURE: host-array path (cfg2 x 16, %s I/O) differs from the C oracle" % label)
                hs = eng.host_stats()
                rec["at_2^20"][label] = {"value": round(big / dt, 1), "ms_per_step": round(dt * 1e3, 3), "best_ms": round(best * 1e3, 3), "chunks": hs["chunks"],
                                         "wall_gbs_both_directions": round(big * nbytes / dt / 1e9, 2),
                                         "planned_kernel_ns_per_elem": round(hs["planned_kernel_ns_per_elem"], 3), "planned_from_measurement": bool(hs["planned_from_measurement"])}
            # The same raw-R1 call on the constant-time context: its chunks are planned with ITS kernels' measured time.
            # floor = the constant-time kernels device-resident for 2^20 + one generation's copy in + copy out at the link's measured rates.
            eng_ct = getattr(self, "eng_ct", None)
            if eng_ct is not None and ct_ms:
                dt, best, got = self.timed_calls(lambda: eng_ct.mul_endo(sb, pb, out=ob), reps_big)
                if not all(np.array_equal(got[k * n:(k + 1) * n], want_words) for k in range(16)):
                    raise SystemExit("PARITY FAILURE: host-array path (cfg2 x 16, raw R1, constant-time context) differs from the C oracle")
                hs = eng_ct.host_stats()
                floor = 16 * ct_ms + n * 192 / (rec["gbs_h2d"] * 1e6) + n * 160 / (rec["gbs_d2h"] * 1e6)
                rec["at_2^20"]["ct_r1"] = {"value": round(big / dt, 1), "ms_per_step": round(dt * 1e3, 3), "best_ms": round(best * 1e3, 3), "chunks": hs["chunks"],
                                           "floor_ms": round(floor, 3), "over_floor": round(dt * 1e3 / floor, 3),
                                           "planned_kernel_ns_per_elem": round(hs["planned_kernel_ns_per_elem"], 3), "planned_from_measurement": bool(hs["planned_from_measurement"])}
        for a in pins:
            eng.host_free(a)
        return rec

    def ct_select_record(self, workload, n, steps, warmup, want_words, default_ms):
        """The same workload on a SECOND context with constant-time table selection (fourq_ctx_set_ct_select): every ladder step
        reads the whole table and keeps the wanted entry by masks.  Outputs must equal the default mode's oracle-checked ones."""
        import numpy as np
        import torch
        from fourq_amd import Engine
        if getattr(self, "eng_ct", None) is None:
            self.eng_ct = GPU.engine(self.local_rank, self.stream)
            self.eng_ct.ct_select = True
        default, self.eng = self.eng, self.eng_ct
        try:
            rec, d = self.run(workload, n, steps, warmup)
        finally:
            self.eng = default
        ok = want_words is not None and bool(np.array_equal(d["out"].cpu().numpy().view(np.uint64), want_words))
        if want_words is not None and not ok:
            raise SystemExit("PARITY FAILURE: constant-time selection mode differs from the C oracle (%s)" % workload)
        del d
        GPU.empty_cache()
        return {"ms_per_step": rec["ms_per_step"], "value": rec["value"], "unit": rec["unit"], "steps": steps, "batch_per_gpu": n,
                "ratio_vs_default": round(rec["ms_per_step"] / default_ms, 3), "parity_ok": ok if want_words is not None else None}


def _pmc_traffic(workload, build_id):
    """(HBM-side bytes per bench step, where that figure comes from).  The bytes are PMC counter readings of a profile run
    committed under profiles/ (tools/profile_all.sh), not a property of this run: they are reported only when that profile was
    taken on the library build that is loaded now, otherwise traffic is None and the source says why."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        with open(path) as fh:
            data = json.load(fh)
    except (OSError, ValueError):
        return None, {"file": None, "note": "profiles/pmc_traffic.json not found"}
    profiled = (data.get("library_build_ids") or {}).get(workload, data.get("library_build_id"))
    src = {"file": "profiles/pmc_traffic.json <- %s" % data.get("source"), "profiled_library_build_id": profiled, "loaded_library_build_id": build_id}
    if not build_id or profiled != build_id:
        src["claimed"] = False          # the committed PMC profile was taken on another build of the library: no traffic figure for this one
        return None, src
    return (data.get("per_workload") or {}).get(workload), src


def _pmc_valu_instructions(workload, build_id):
    """wave64 VALU instructions per bench step (SQ_INSTS_VALU summed over the step's kernels) from the same committed profile, under the
    same rule: only for the build that is loaded."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as fh:
            data = json.load(fh)
    except (OSError, ValueError):
        return None
    profiled = (data.get("library_build_ids") or {}).get(workload, data.get("library_build_id"))
    if not build_id or profiled != build_id:
        return None
    return (data.get("valu_wave_instructions_per_step") or {}).get(workload)


def _built_from_sources():
    """True when the loaded library's compiled-in id equals the hash of the sources beside it (fourq_amd._lib.build_matches_sources)."""
    try:
        from fourq_amd import _lib
        return _lib.build_matches_sources()
    except Exception:
        return None


# ---- what is printed ---------------------------------------------------------------------------------------------
LINE_MAX_BYTES = 6144        # the driver reads the tail of stdout: the line must stay far below what it keeps (22 KB was not parsed)


def _get(d, *path):
    for k in path:
        if not isinstance(d, dict) or k not in d:
            return None
        d = d[k]
    return d


def compact_config(rec, ct=None):
    """One flat, numbers-only record (<= 300 bytes) of a workload's full record."""
    out = {"value": rec["value"], "unit": rec["unit"], "ms_per_step": rec["ms_per_step"], "kernel_ms": _get(rec, "roofline", "kernel_ms"),
           "roofline_frac": _get(rec, "roofline", "frac"), "algorithmic_frac": _get(rec, "valu_roofline", "algorithmic_frac"),
           "issue_frac": _get(rec, "valu_roofline", "issue", "frac"), "cycles_per_unit": rec.get("cycles_per_unit"),
           "parity_ok": _get(rec, "parity", "ok"), "pcie_value": _get(rec, "pcie_inclusive", "value"),
           "ct_ratio": None if not ct else ct.get("ratio_vs_default")}
    if rec.get("gather_ms") is not None:
        out["gather_ms"] = rec["gather_ms"]
    return out


def compact_line(full):
    """The ONE line of stdout: the contract keys and numbers only, <= LINE_MAX_BYTES.  Every field is defined in profiles/BENCH_FIELDS.md;
    the full record (every nested measurement) goes to stderr and to bench_full.json."""
    cfg, roof, valu = full["config"], full["roofline"], full["valu_roofline"]
    ct = full.get("ct_select") or {}
    head = next((w for w in WORKLOADS if WORKLOADS[w]["text"] == cfg["workload"]), "cfg2")
    line = {k: full[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")}
    line["config"] = {"workload": cfg["workload"][:120], "batch_per_gpu": cfg["batch_per_gpu"], "ranks_seen": cfg["ranks_seen"], "backend": cfg["backend"],
                      "build_id": _get(cfg, "library", "build_id"), "built_from_sources": _get(cfg, "library", "built_from_these_sources"),
                      "table_selection": cfg["table_selection"]}
    line["roofline"] = {k: roof.get(k) for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "kernel_ms", "algorithmic_bytes_per_launch")}
    line["valu_roofline"] = {"peak": valu["peak"], "unit": valu["unit"], "algorithmic_frac": valu["algorithmic_frac"], "executed_frac": valu["executed_frac"],
                             "issue_frac": _get(valu, "issue", "frac")}
    line["clock_mhz"] = _get(full, "clock", "in_kernel_mhz")
    line["cycles_per_unit"] = full.get("cycles_per_unit")
    line["parity_ok"] = _get(full, "parity", "all_ranks_ok") if full["n_gpus"] > 1 else _get(full, "parity", "ok")
    if "cycles_per_unit_ranks" in full:
        line["cycles_per_unit_ranks"] = full["cycles_per_unit_ranks"]
    if "gather_ms" in full:
        line["gather_ms"] = full["gather_ms"]
    cb = full.get("cpu_baseline")
    if cb:
        line["cpu_baseline"] = {"value": cb["value"], "unit": cb["unit"], "cores": cb["cores"], "host_cores_total": cb["host_cores_total"],
                                "host_cores_granted": cb["host_cores_granted"], "cores_cap": cb["cores_cap"], "kind": cb["kind"], "per_core": cb["per_core"],
                                "per_core_reference_survey": cb["per_core_reference_survey"], "c_restatement_value": _get(cb, "c_restatement", "value"),
                                "sample": cb["sample"][:160]}
    if head in ct:
        line["ct_value"], line["ct_ratio"] = ct[head]["value"], ct[head]["ratio_vs_default"]
    if "alongside" in full:
        line["alongside_value"] = full["alongside"]["value_per_gpu"]
    p = full.get("pcie_inclusive")
    if p:
        big = p.get("at_2^20") or {}
        line["pcie"] = {"value": p["value"], "ms": p["ms_per_step"], "pageable_ms": _get(p, "pageable_caller", "ms_per_step"),
                        "affine_ms": _get(p, "affine", "ms_per_step"), "bytes_ms": _get(p, "bytes", "ms_per_step"),
                        "r1_2p20_ms": _get(big, "r1", "ms_per_step"), "affine_2p20_ms": _get(big, "affine", "ms_per_step"), "bytes_2p20_ms": _get(big, "bytes", "ms_per_step"),
                        "ct_r1_2p20_ms": _get(big, "ct_r1", "ms_per_step"), "ct_floor_2p20_ms": _get(big, "ct_r1", "floor_ms")}
    if "configs" in full:
        line["configs"] = {w: compact_config(r, ct.get(w)) for w, r in sorted(full["configs"].items())}
    line["full_record"] = full.get("full_record")
    line["fields"] = "profiles/BENCH_FIELDS.md"
    text = json.dumps(line, separators=(",", ":"))
    if len(text) > LINE_MAX_BYTES:
        raise SystemExit("bench.py: the result line is %d bytes, over the %d the driver is known to read" % (len(text), LINE_MAX_BYTES))
    return text


def emit(full, full_json):
    """Full record -> stderr and `full_json`; then the compact line, alone, on stdout."""
    full["full_record"] = os.path.relpath(full_json, ROOT) if full_json else None
    blob = json.dumps(full)
    if full_json:
        try:
            with open(full_json, "w") as fh:
                fh.write(blob + "\n")
        except OSError as exc:                      # a read-only tree: the record is still on stderr
            full["full_record"] = None
            print("bench.py: could not write %s (%s)" % (full_json, exc), file=sys.stderr)
    print(blob, file=sys.stderr, flush=True)
    text = compact_line(full)
    sys.stdout.flush()
    if _REAL_STDOUT is None:
        print(text, flush=True)
    else:
        os.write(_REAL_STDOUT, (text + "\n").encode())


_REAL_STDOUT = None


def stdout_is_for_the_line_only():
    """From here on file descriptor 1 of this process IS its stderr: whatever a library prints on stdout (gloo's connection banner, a
    runtime's notices) cannot end up beside the result line.  emit() writes the line to the real stdout kept aside here."""
    global _REAL_STDOUT
    if _REAL_STDOUT is None:
        sys.stdout.flush()
        _REAL_STDOUT = os.dup(1)
        os.dup2(2, 1)


# ---- launcher ----------------------------------------------------------------------------------------------------
def self_launch(args, argv):
    """`bench.py --gpus N` from a bare shell: N fresh rank processes under torch.distributed.run.  This process never
    imports torch.cuda or creates an engine, and nothing that has touched a GPU is re-exec'd."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    # The pool's host driver supports dmabuf IPC only: without this RCCL's (and torch's) cross-process buffer sharing fails with
    # "hipIpcGetMemHandle: invalid argument".  The GPU boxes export it already (the build environment's own note); a bare shell may not.
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, min(host_cores(), host_cores(cap=False) // args.gpus))))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + argv
    # stdout carries rank 0's JSON line and nothing else: whatever the ranks' libraries print there (gloo's connection
    # banner in rehearsals) is passed on to stderr
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    for line in proc.stdout.splitlines():
        print(line, file=sys.stdout if line.startswith('{"metric"') else sys.stderr, flush=True)
    return proc.returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=0, help="timed steps of the headline workload (default 500 for cfg2; the workload's own default otherwise)")
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="cfg2", help="BASELINE.json configuration (default: the headline, cfg2)")
    ap.add_argument("--batch", type=int, default=0, help="units per GPU per step (default: the workload's BASELINE size)")
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the pure-Python CPU baseline leg (rank 0)")
    ap.add_argument("--no-configs", action="store_true", help="headline only: skip the nested cfg3/cfg4/cfg5 records")
    ap.add_argument("--no-pcie", action="store_true", help="skip the PCIe-inclusive host-array measurement")
    ap.add_argument("--no-alongside", action="store_true", help="skip the second operation SURVEY.md 8(d) reports alongside cfg2 / cfg3")
    ap.add_argument("--no-parity", action="store_true", help="profiling runs only: skip the whole-shard C-oracle gate")
    ap.add_argument("--full-json", default=os.path.join(ROOT, "bench_full.json"), help="where the full record goes (also printed on stderr); '' = nowhere")
    ap.add_argument("--no-ct", action="store_true", help="skip the constant-time-selection records (second context, FOURQ_CT_SELECT mode)")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(self_launch(args, sys.argv[1:]))

    stdout_is_for_the_line_only()
    import numpy as np
    import torch
    import torch.distributed as dist
    from fourq_amd import codec
    from fourq_amd.dist import env_rank

    # FOURQ_BENCH_REHEARSE=1: every rank on GPU 0 over gloo -- exercises the N>1 code path on a one-GPU box
    rehearse = os.environ.get("FOURQ_BENCH_REHEARSE") == "1"
    rank, local_rank, world = env_rank()
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if rehearse:
        local_rank = 0
    GPU.set_device(local_rank)
    dist_on = world > 1 or os.environ.get("FOURQ_BENCH_FORCE_DIST") == "1"
    if dist_on:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        import datetime
        # a rank that dies (its parity gate raises) must take the job down, not leave the others waiting for ever
        dist.init_process_group(backend="gloo" if rehearse else "nccl", rank=rank, world_size=world,
                                timeout=datetime.timedelta(seconds=int(os.environ.get("FOURQ_BENCH_PG_TIMEOUT_S", "600"))))
        # RCCL builds its communicator at the FIRST collective (hundreds of milliseconds, the GPU idle meanwhile).  Left to the barrier in
        # front of the timed steps, that wait let the clock governor fall back right before the measurement: 20 timed steps at 1 891 MHz
        # instead of 2 365 -- 1.67 x 10^8/s instead of 2.2 (profiles/r06_rccl_group_of_one.txt).  So: every collective the rank body uses,
        # once, here, before anything is timed.
        where = "cpu" if rehearse else torch.device("cuda", local_rank)
        dist.barrier()
        for op in (dist.ReduceOp.MAX, dist.ReduceOp.MIN, dist.ReduceOp.SUM):
            dist.all_reduce(torch.zeros(1, dtype=torch.float64, device=where), op=op)
        from fourq_amd.dist import gather_rows
        gather_rows(torch.zeros((16, 8), dtype=torch.int64, device=where), 16 * world, dst=0)
        GPU.synchronize()
    b = Bench(rank, local_rank, world, rehearse, dist_on)

    wl = WORKLOADS[args.workload]
    n = args.batch or wl["batch"]
    steps = args.steps or wl["steps"]
    rec, d = b.run(args.workload, n, steps, args.warmup)
    parity, got, want = ({"gate": "skipped (--no-parity)", "ok": None}, None, None) if args.no_parity else b.parity_gate(args.workload, d)
    if not args.no_parity:                                  # the edge-case mini-batch belongs to EVERY run, on every rank (raises on a difference)
        parity["edge_cases_checked"] = edge_case_check(b.eng)
    gather = b.gather_ms(d["out"], n) if dist_on else None
    alongside = None
    if rank == 0 and args.workload in ("cfg2", "cfg3") and not args.no_alongside and not args.no_parity:
        alongside = b.alongside(args.workload, d, steps=100 if args.workload == "cfg2" else 20)

    # constant-time selection mode beside the default, on a second context (one GPU only: the N > 1 runs measure scaling)
    do_ct = world == 1 and not args.no_ct and not b.eng.ct_select
    ct = {}
    if do_ct:
        ct[args.workload] = b.ct_select_record(args.workload, n, max(5, steps // 5), 2, want, rec["ms_per_step"])

    configs = {}
    others = [] if (args.no_configs or args.batch) else [w for w in sorted(WORKLOADS) if w != args.workload]
    pcie_reps = {"cfg2": 21, "cfg3": 5, "cfg4": 5, "cfg5": 11}
    for w in others:
        r, dw = b.run(w, WORKLOADS[w]["batch"], WORKLOADS[w]["steps"], max(2, args.warmup // 4))
        if args.no_parity:
            r["parity"], want_w = {"gate": "skipped (--no-parity)", "ok": None}, None
        else:
            r["parity"], _, want_w = b.parity_gate(w, dw)
        if dist_on and w == "cfg4":
            r["gather_ms"] = b.gather_ms(dw["out"], WORKLOADS[w]["batch"])
        if world == 1 and not args.no_pcie and want_w is not None:
            r["pcie_inclusive"] = b.pcie_inclusive(w, dw, want_w, reps=pcie_reps[w])
        if rank == 0 and w == "cfg3" and not args.no_alongside and not args.no_parity:
            r["alongside"] = b.alongside("cfg3", dw, steps=20)
        configs[w] = r
        del dw
        GPU.empty_cache()
        if do_ct:
            ct[w] = b.ct_select_record(w, WORKLOADS[w]["batch"], max(5, WORKLOADS[w]["steps"] // 5), 2, want_w, r["ms_per_step"])

    ranks_seen, cyc_range = 1, None
    if dist_on:                                             # every rank passed its own gate, or the job has already died
        where = "cpu" if rehearse else b.dev
        ok = torch.tensor([1.0 if parity["ok"] in (True, None) else 0.0], device=where)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        parity["all_ranks_ok"] = bool(ok.item() == 1.0)
        # ranks_seen is a record, not a guard: a rank that died makes the collectives above fail or time out (init_process_group's
        # timeout), and this process then exits non-zero without a line (tests/test_dist.py)
        here = torch.tensor([1.0], device=where)
        dist.all_reduce(here, op=dist.ReduceOp.SUM)
        ranks_seen = int(round(here.item()))
        if ranks_seen != args.gpus or not parity["all_ranks_ok"]:
            raise SystemExit("bench.py --gpus %d: %d ranks reported, all_ranks_ok=%s -- no result line is printed for an incomplete job"
                             % (args.gpus, ranks_seen, parity["all_ranks_ok"]))
        cyc = rec["cycles_per_unit"] or 0.0                                     # per-rank cycles per unit: the spread over the node's devices
        lo, hi = torch.tensor([cyc if cyc else 1e30], dtype=torch.float64, device=where), torch.tensor([cyc], dtype=torch.float64, device=where)
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        cyc_range = [round(float(lo.item()), 3) if lo.item() < 1e29 else None, round(float(hi.item()), 3) or None]
    if rank == 0:
        full = {
            "metric": "FourQ scalar-mults/sec (batch, whole node)", "value": rec["value"], "unit": rec["unit"],
            "n_gpus": world, "steps": steps, "warmup": args.warmup, "ms_per_step": rec["ms_per_step"],
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
            "config": {"workload": rec["workload"], "batch_per_gpu": n, "parallelism": "independent shards x%d, no data-path collective" % world,
                       "clock_settle_ms": b.settle_ms, "table_selection": "constant-time" if b.eng.ct_select else "indexed",
                       "library": {"version": b.eng.version, "build_id": b.eng.build_id, "built_from_these_sources": _built_from_sources()},
                       "ranks_seen": ranks_seen,
                       "backend": ("gloo (rehearsal: every rank on GPU 0)" if rehearse else "nccl (RCCL)") if dist_on else None},
            "roofline": rec["roofline"], "valu_roofline": rec["valu_roofline"], "parity": parity,
            "clock": rec["clock"], "cycles_per_unit": rec["cycles_per_unit"], "kernel_cycles_per_step": rec["kernel_cycles_per_step"],
            "fields": "profiles/BENCH_FIELDS.md",
        }
        if cyc_range is not None:
            full["cycles_per_unit_ranks"] = cyc_range
        if gather is not None:
            full["gather_ms"] = gather
        if alongside is not None:
            full["alongside"] = alongside
        if configs:
            full["configs"] = configs
        if world == 1 and not args.no_pcie and want is not None:
            full["pcie_inclusive"] = b.pcie_inclusive(args.workload, d, want, reps=pcie_reps[args.workload], ct_ms=(ct.get(args.workload) or {}).get("ms_per_step"))
        if world == 1 and args.workload == "cfg2" and not args.batch and not args.no_alongside:
            full["size_sweep"] = b.size_sweep(d)
            if not args.no_parity:
                full["small_batches"] = b.small_batches(d)
        if ct:
            full["ct_select"] = ct
        # The CPU baseline is an N = 1 figure (rank 0, alone on the host): at N > 1 the other ranks' oracle gates would share its cores
        if world == 1 and not args.no_cpu_baseline and got is not None:
            sys.path.insert(0, os.path.join(ROOT, "oracle"))
            sample = min(n, 1 << 16)
            items, extra, what = cpu_items(args.workload, d, sample)
            full["cpu_baseline"] = cpu_baseline(args.workload, items, extra, codec.unpack_points(got[:sample]), what)
            full["cpu_baseline"]["c_restatement"] = {
                "value": parity["c_oracle_units_per_s"], "unit": "scalar-mults/s", "threads": parity["c_oracle_threads"], "cores": host_cores(), "kind": "port",
                "sample": "whole batch (%d units) via oracle/fourq_oracle.c (OpenMP), every output compared bit-exact with the GPU's" % n}
            full["cpu_baseline"]["edge_cases_checked"] = parity.get("edge_cases_checked")
        emit(full, args.full_json)
    if getattr(b, "eng_ct", None) is not None:
        b.eng_ct.close()
    b.eng.close()
    if dist_on:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
