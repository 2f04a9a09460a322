"""SURVEY.md 8(e) on hardware: the engine's N > 1 paths.

* `MultiEngine` (one context and one host thread per device, contiguous shards, no collective) gives the results of the
  C oracle / of a single `Engine` for every host-array entry point.  A one-GPU box runs it as two contexts on device 0.
* `bench.py --gpus 2` in rehearsal mode (both ranks on GPU 0 over gloo) runs the launcher, the per-rank shard, the
  barrier-bracketed timed region, the whole-shard C-oracle gate on every rank and the gather -- the code the driver's
  8-GPU run executes with RCCL in place of gloo.
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", params=[False, True], ids=["select-by-address", "constant-time"])
def pair(request):
    from fourq_amd import Engine, MultiEngine
    multi, single = MultiEngine([0, 0]), Engine(0)
    multi.ct_select = single.ct_select = request.param
    yield multi, single
    multi.close()
    single.close()


def _inputs(eng, n, seed):
    from bench import seeded_scalars
    from fourq_amd import codec, constants
    g1 = codec.pack_point((constants.Gx, constants.Gy, (1, 0), constants.Gx, constants.Gy))
    table = eng.table_endo(g1)
    s, k = seeded_scalars(seed, n), seeded_scalars(seed + 1, n)
    return g1, table, s, k


def test_device_count_and_shape(pair):
    from fourq_amd import device_count
    multi, single = pair
    assert device_count() >= 1 and len(multi.engines) == 2 and multi.lanes == 2 * single.lanes


def test_mul_endo_two_contexts_vs_c_oracle(pair):
    """2^17 + 1 variable-base elements: both shards take more than one fused generation and differ in size by one."""
    import oracle_c as oc
    multi, single = pair
    n = (1 << 17) + 1
    g1, table, s, k = _inputs(single, n, 91)
    pts = multi.mul_endo_fixed(k, table)
    assert np.array_equal(pts, oc.mul(oc.ENDO, k, None, table))
    got = multi.mul_endo(s, pts)
    assert np.array_equal(got, oc.mul(oc.ENDO, s, pts))


def test_every_entry_point_matches_the_single_engine(pair):
    from fourq_amd import codec, constants
    multi, single = pair
    n = 4099                                                     # odd: the two shards differ by one row
    g1, table, s, k = _inputs(single, n, 17)
    pts = single.mul_endo_fixed(k, table)
    assert np.array_equal(multi.mul_windowed(s, pts), single.mul_windowed(s, pts))
    tw = single.table_windowed(g1)
    assert np.array_equal(multi.mul_windowed_fixed(s, tw), single.mul_windowed_fixed(s, tw))
    flags = (k[:, 0] & 1).astype(np.uint8)
    assert np.array_equal(multi.mul_endo_mixed(s, pts, flags, table), single.mul_endo_mixed(s, pts, flags, table))
    gaff = np.repeat(codec.pack_point((constants.Gx, constants.Gy)).reshape(1, 8), n, axis=0)
    gaff[5, 0] ^= 1                                              # one point off the curve: status 1 in the first shard
    for a, b in zip(multi.dh_endo(k, gaff), single.dh_endo(k, gaff)):
        assert np.array_equal(a, b)
    pub, st = single.dh_endo(k, gaff)
    assert st[5] == 1 and not st[6:].any()
    for a, b in zip(multi.dh_windowed(s, pub), single.dh_windowed(s, pub)):
        assert np.array_equal(a, b)
    for a, b in zip(multi.dh_exchange(s, k, gaff[0]), single.dh_exchange(s, k, gaff[0])):
        assert np.array_equal(a, b)
    g392 = single.mul_endo(codec.pack_scalars([392]), g1.reshape(1, 20))[0]
    comb = single.comb_table(g392)
    for a, b in zip(multi.comb_mul(k, comb), single.comb_mul(k, comb)):
        assert np.array_equal(a, b)
    enc = single.encode(pub[6:])
    assert np.array_equal(multi.encode(pub[6:]), enc)
    for a, b in zip(multi.decode(enc), single.decode(enc)):
        assert np.array_equal(a, b)
    for a, b in zip(multi.dh_bytes(s[6:], enc), single.dh_bytes(s[6:], enc)):
        assert np.array_equal(a, b)


def test_tiny_and_empty_batches(pair):
    multi, single = pair
    g1, table, s, k = _inputs(single, 3, 5)
    assert np.array_equal(multi.mul_endo_fixed(s[:1], table), single.mul_endo_fixed(s[:1], table))      # one row: second shard empty
    assert multi.mul_endo_fixed(s[:0], table).shape == (0, 20)
    pts = single.mul_endo_fixed(k, table)
    assert np.array_equal(multi.mul_endo(s, pts), single.mul_endo(s, pts))
    with pytest.raises(ValueError):
        multi.mul_endo(s, pts[:2])


def test_bench_two_ranks_rehearsal_on_one_gpu():
    """`bench.py --gpus 2` from a bare shell with FOURQ_BENCH_REHEARSE=1: launcher -> two ranks on GPU 0 over gloo."""
    env = dict(os.environ, FOURQ_BENCH_REHEARSE="1", FOURQ_BENCH_SETTLE_MS="10")
    for var in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(var, None)
    proc = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "cfg2", "--steps", "5", "--warmup", "1",
                           "--no-cpu-baseline", "--no-configs"], capture_output=True, text=True, env=env, timeout=900)
    assert proc.returncode == 0, proc.stderr[-3000:]
    lines = [ln for ln in proc.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines[:3]
    assert len(lines[0]) <= 6144
    line = json.loads(lines[0])
    full = json.loads([ln for ln in proc.stderr.splitlines() if ln.startswith('{"metric"')][-1])        # rank 0's full record, passed through on stderr
    assert line["n_gpus"] == 2 and line["config"]["ranks_seen"] == 2 and "gloo" in line["config"]["backend"]
    assert line["parity_ok"] is True and full["parity"]["ok"] is True and full["parity"]["all_ranks_ok"] is True and full["parity"]["units"] == 1 << 16
    assert line["gather_ms"] > 0 and line["scaling"] == "weak" and "cpu_baseline" not in line
    assert abs(line["value"] - 2 * (1 << 16) / (line["ms_per_step"] * 1e-3)) / line["value"] < 0.01     # whole-job units / max-over-ranks time
    assert full["parity"]["c_oracle_threads"] >= 1                                                      # each rank took a share of the cores
    lo, hi = line["cycles_per_unit_ranks"]                                                              # two ranks time-sharing one GPU: about twice a lone rank's
    assert 8.0 < lo <= hi < 40.0


def test_one_context_called_from_several_threads():
    """The reference's functions are pure, so a drop-in caller may use them from any thread; `fourq_amd.curve4q` runs them all on one
    process-wide context.  Calls on one context take turns under the context's lock (include/fourq_amd.h, "Threads"): eight threads
    mixing single reference-shaped calls, small and pipelined host-array batches, DH and device-pointer calls on ONE engine, every
    result compared with the C oracle.  (ctypes drops the GIL for the duration of a call, so the calls do overlap in time.)"""
    import threading
    import torch
    import oracle_c as oc
    from bench import seeded_scalars
    from fourq_amd import Engine, codec, constants, curve4q
    from fourq_amd.engine import default_engine
    eng = default_engine()
    g1 = codec.pack_point((constants.Gx, constants.Gy, (1, 0), constants.Gx, constants.Gy))
    te = oc.table(oc.ENDO, g1)
    G = codec.pack_point((constants.Gx, constants.Gy))
    sizes = [1, 3, 700, 5000, eng.lanes // 2 + 5, eng.lanes + 17, 2 * eng.lanes + 9, 40]
    errors = []

    def work(t):
        try:
            n = sizes[t]
            for rep in range(3):
                s, k = seeded_scalars(1000 + 10 * t + rep, n), seeded_scalars(2000 + 10 * t + rep, n)
                pts = eng.mul_endo_fixed(k, te)
                assert np.array_equal(pts, oc.mul(oc.ENDO, k, None, te)), "fixed"
                assert np.array_equal(eng.mul_endo(s, pts), oc.mul(oc.ENDO, s, pts)), "variable"
                m = min(n, 3000)
                g = np.repeat(G.reshape(1, 8), m, axis=0)
                out, st = eng.dh_endo(s[:m], g)
                want, ws = oc.dh(oc.ENDO, s[:m], g)
                assert np.array_equal(out, want) and np.array_equal(st, ws), "dh"
                if t % 2:                                              # device-pointer calls from this thread, on the shared context's stream
                    dev = torch.device("cuda", 0)
                    s_d, p_d = torch.from_numpy(s.view(np.int64)).to(dev), torch.from_numpy(pts.view(np.int64)).to(dev)
                    o_d = torch.empty((n, 20), dtype=torch.int64, device=dev)
                    torch.cuda.synchronize()
                    eng.mul_windowed_dev(s_d, p_d, o_d, n)
                    eng.sync()
                    assert np.array_equal(o_d.cpu().numpy().view(np.uint64), oc.mul(oc.WINDOWED, s, pts)), "dev"
                mi = int.from_bytes(s[0].tobytes(), "little")
                P = codec.unpack_fp2s(pts[0])
                assert curve4q.MUL_endo(mi, P) == codec.unpack_fp2s(oc.mul(oc.ENDO, s[:1], pts[:1])[0]), "single"
        except BaseException as e:                                     # noqa: BLE001 -- reported by the main thread
            errors.append((t, repr(e)))

    threads = [threading.Thread(target=work, args=(t,)) for t in range(len(sizes))]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors


def test_concurrent_single_calls_are_combined():
    """`curve4q.MUL_endo(m, P)` etc. from 16 threads: the calls that arrive while a batch is on the GPU leave as one batch
    (fourq_amd/combine.py); every caller gets its own result -- and its own exception: one thread's DH calls are made on a point
    that is not on the curve and raise there, and only there."""
    import threading
    import oracle_c as oc
    from bench import seeded_scalars
    from fourq_amd import codec, constants, curve4q
    T, R = 16, 25
    g1 = codec.pack_point((constants.Gx, constants.Gy, (1, 0), constants.Gx, constants.Gy))
    te = oc.table(oc.ENDO, g1)
    s, k = seeded_scalars(3001, T * R), seeded_scalars(3002, T * R)
    pts = oc.mul(oc.ENDO, k, None, te)
    want_e, want_w = oc.mul(oc.ENDO, s, pts), oc.mul(oc.WINDOWED, s, pts)
    from fourq_amd.engine import default_engine
    aff = default_engine().prim("PT_R1TOAFFINE", pts)             # DH on the points themselves (DH on the generator is key generation: the comb)
    want_dh, st = oc.dh(oc.ENDO, s, aff)
    assert not st.any()
    g = np.repeat(codec.pack_point((constants.Gx, constants.Gy)).reshape(1, 8), T * R, axis=0)
    want_kg, st = oc.dh(oc.ENDO, s, g)
    assert not st.any()
    ints = [int.from_bytes(row.tobytes(), "little") for row in s]
    tuples = [codec.unpack_fp2s(row) for row in pts]
    affs = [codec.unpack_fp2s(row) for row in aff]
    before = {kind: v["calls"] for kind, v in curve4q.combine_stats().items()}
    errors = []

    def work(t):
        try:
            for r in range(R):
                i = t * R + r
                assert curve4q.MUL_endo(ints[i], tuples[i]) == codec.unpack_fp2s(want_e[i]), "MUL_endo"
                if r % 5 == 0:
                    assert curve4q.MUL_windowed(ints[i], tuples[i]) == codec.unpack_fp2s(want_w[i]), "MUL_windowed"
                if t == 3:
                    with pytest.raises(Exception, match="Point not on curve"):
                        curve4q.DH_endo(ints[i], ((1, 2), (3, 4)))
                else:
                    assert curve4q.DH_endo(ints[i], affs[i]) == codec.unpack_fp2s(want_dh[i]), "DH_endo"
                    if r % 5 == 1:
                        assert curve4q.DH_endo(ints[i], (constants.Gx, constants.Gy)) == codec.unpack_fp2s(want_kg[i]), "keygen"
        except BaseException as e:                                     # noqa: BLE001 -- reported by the main thread
            errors.append((t, repr(e)))

    threads = [threading.Thread(target=work, args=(t,)) for t in range(T)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors
    stats = curve4q.combine_stats()
    assert stats["mul_endo"]["calls"] - before.get("mul_endo", 0) == T * R and stats["dh_endo"]["calls"] - before.get("dh_endo", 0) == T * R
    assert stats["mul_endo"]["batches"] < stats["mul_endo"]["calls"] and stats["mul_endo"]["largest_batch"] > 1
    assert stats["keygen_comb"]["calls"] - before.get("keygen_comb", 0) == (T - 1) * (R // 5)


def test_rccl_loads_and_runs_collectives_on_this_image():
    """The pool gives this build one GPU, so no multi-rank RCCL job can run here -- but the backend bench.py --gpus N uses ("nccl" IS RCCL
    on ROCm) can at least be initialised and driven through the calls the rank body makes: barrier, all_reduce(MAX / MIN / SUM) on the
    tensors bench.py reduces, and the gather of fourq_amd.dist.gather_rows' padded buffers, on a process group of ONE rank in a fresh
    process (a group of one still loads librccl, creates a communicator on the device and enqueues its kernels on the stream)."""
    import subprocess
    import sys
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    code = (
        "import datetime, torch, torch.distributed as dist\n"
        "torch.cuda.set_device(0)\n"
        "dist.init_process_group('nccl', rank=0, world_size=1, timeout=datetime.timedelta(seconds=120))\n"
        "dev = torch.device('cuda', 0)\n"
        "dist.barrier()\n"
        "t = torch.tensor([1.25], dtype=torch.float64, device=dev)\n"
        "for op in (dist.ReduceOp.MAX, dist.ReduceOp.MIN, dist.ReduceOp.SUM):\n"
        "    dist.all_reduce(t, op=op)\n"
        "assert float(t.item()) == 1.25\n"
        "pad = torch.arange(8 * 524288, dtype=torch.int64, device=dev).reshape(524288, 8)      # config 4's shard: 2^19 affine rows\n"
        "bufs = [torch.empty_like(pad)]\n"
        "dist.gather(pad, bufs, dst=0)\n"
        "torch.cuda.synchronize()\n"
        "assert torch.equal(bufs[0], pad)\n"
        "print('backend', dist.get_backend(), 'ok')\n"
        "dist.destroy_process_group()\n")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    for var in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(var, None)
    proc = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=300)
    assert proc.returncode == 0, proc.stderr[-3000:]
    assert "backend nccl ok" in proc.stdout


def test_bench_rank_body_over_rccl_with_a_group_of_one():
    """bench.py's `nccl` branch had never executed.  FOURQ_BENCH_FORCE_DIST=1 makes the UNMODIFIED rank body create its
    process group and run every collective it runs at N > 1 -- the barriers around the timed steps, the MAX over ranks, the MIN / SUM of
    the parity verdicts, the per-rank cycles' MIN / MAX, the gather of results (fourq_amd.dist.gather_rows) -- over RCCL, on a group of
    one rank: all that a one-GPU box can offer, and it is the same code path an 8-GPU job takes."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, FOURQ_BENCH_FORCE_DIST="1", FOURQ_BENCH_SETTLE_MS="10", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
               RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("FOURQ_BENCH_REHEARSE", None)
    proc = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "5", "--warmup", "1", "--no-cpu-baseline", "--no-pcie", "--no-ct",
                           "--no-alongside", "--full-json", ""], capture_output=True, text=True, env=env, timeout=900)
    assert proc.returncode == 0, proc.stderr[-3000:]
    lines = [ln for ln in proc.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1 and len(lines[0]) <= 6144
    line = json.loads(lines[0])
    assert line["config"]["backend"] == "nccl (RCCL)" and line["config"]["ranks_seen"] == 1 and line["n_gpus"] == 1
    assert line["parity_ok"] is True and line["gather_ms"] > 0
    lo, hi = line["cycles_per_unit_ranks"]
    assert lo == hi == line["cycles_per_unit"]
    assert line["configs"]["cfg4"]["gather_ms"] > 0 and all(c["parity_ok"] for c in line["configs"].values())
