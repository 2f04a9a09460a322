"""The N>1 path on CPU: two processes over gloo shard a batch contiguously, each computes its
shard, and rank 0 gathers.  On the GPU box the same code runs with backend "nccl" (RCCL) and the
HIP engine; here the per-shard compute is the C oracle so that the test needs no GPU."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT
from fourq_amd.dist import gather_rows, shard_bounds, sharded_map


def test_shard_bounds_cover_without_overlap():
    for n in (0, 1, 7, 64, 65, 1 << 16, (1 << 22) + 3):
        for world in (1, 2, 3, 4, 8):
            cuts = [shard_bounds(n, r, world) for r in range(world)]
            assert cuts[0][0] == 0 and cuts[-1][1] == n
            assert all(cuts[i][1] == cuts[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in cuts]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_bounds(10, 2, 2)


def _worker(rank, world, port, n, result_path):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_c as oc
    from bench import seeded_scalars
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        scalars = seeded_scalars(7, n)
        table = oc.table(oc.ENDO, _g1())
        local = sharded_map(lambda s: oc.mul(oc.ENDO, s, None, table), [scalars], n)
        lo, hi = shard_bounds(n, rank, world)
        assert local.shape == (hi - lo, 20)
        full = gather_rows(torch.from_numpy(local.view(np.int64)), n, dst=0)
        if rank == 0:
            np.save(result_path, full.numpy().view(np.uint64))
        else:
            assert full is None
    finally:
        dist.barrier()
        dist.destroy_process_group()


def _g1():
    from fourq_amd import codec, constants
    return codec.pack_point((constants.Gx, constants.Gy, (1, 0), constants.Gx, constants.Gy))


@pytest.mark.parametrize("n", [257, 1000])
def test_two_rank_shard_and_gather_gloo(tmp_path, n):
    import oracle_c as oc
    from bench import seeded_scalars
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "full.npy")
    mp.spawn(_worker, args=(2, port, n, out), nprocs=2, join=True)
    want = oc.mul(oc.ENDO, seeded_scalars(7, n), None, oc.table(oc.ENDO, _g1()))
    assert np.array_equal(np.load(out), want)


def _gather_worker(rank, world, port, n_total, cols, result_path):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lo, hi = shard_bounds(n_total, rank, world)
        rows = torch.arange(lo, hi, dtype=torch.int64).reshape(-1, 1) * 1000003 + torch.arange(cols, dtype=torch.int64).reshape(1, -1)
        full = gather_rows(rows, n_total, dst=0)
        if rank == 0:
            want = torch.arange(n_total, dtype=torch.int64).reshape(-1, 1) * 1000003 + torch.arange(cols, dtype=torch.int64).reshape(1, -1)
            with open(result_path, "w") as fh:
                fh.write("ok" if full.shape == want.shape and torch.equal(full, want) else "MISMATCH %s" % (tuple(full.shape),))
        else:
            assert full is None
    finally:
        dist.barrier()
        dist.destroy_process_group()


@pytest.mark.parametrize("world,extra", [(2, 1), (3, 2)])
def test_gather_at_config_4s_real_shard_size_with_uneven_shards(tmp_path, world, extra):
    """config 4's one collective at its real per-rank size -- 2^19 affine results of 64 bytes per rank, 32 MiB --
    over gloo, with a total that does not divide by the world size (shards differ by a row; every rank pads to the largest).  The same
    function runs over nccl (RCCL) on the GPUs; what is checked here is its indexing and trimming at full size."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "verdict.txt")
    mp.spawn(_gather_worker, args=(world, port, world * (1 << 19) + extra, 8, out), nprocs=world, join=True)
    assert open(out).read() == "ok"


def test_bench_prints_no_line_for_a_job_that_is_not_whole(tmp_path):
    """`bench.py --gpus N` must not print a result line unless N ranks reported and every rank's outputs passed its
    oracle gate.  Two ranks over gloo (tests/bench_cpu_rank.py); rank 1's engine returns one wrong word: its gate raises, the job dies,
    and rank 0 prints nothing that looks like a result."""
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    argv = ["--gpus", "2", "--batch", "256", "--steps", "1", "--warmup", "0", "--no-configs", "--no-pcie", "--no-alongside", "--no-ct", "--no-cpu-baseline"]
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="2",
                   FOURQ_STANDIN_CORRUPT_RANK="1")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "bench_cpu_rank.py")] + argv, env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        try:
            outs.append(p.communicate(timeout=180))
        except subprocess.TimeoutExpired:                     # a rank left waiting for its dead peer: that is a failed job too
            p.kill()
            outs.append(p.communicate())
    assert procs[1].returncode != 0 and "PARITY FAILURE" in outs[1][1]
    assert procs[0].returncode != 0
    assert not [ln for o in outs for ln in o[0].splitlines() if ln.startswith('{"metric"')]


def test_bench_gpus_n_launches_its_own_ranks_before_touching_a_gpu(monkeypatch, capsys):
    """`python bench.py --gpus N` from a bare shell: the parent builds a torch.distributed.run command on 127.0.0.1 with
    a free port, relays the children's return code and never initialises the GPU runtime itself."""
    import types
    import bench
    seen = {}

    def fake_run(cmd, env=None, stdout=None, text=None):
        seen["cmd"], seen["env"] = cmd, env
        return types.SimpleNamespace(returncode=7, stdout='[Gloo] Rank 0 is connected\n{"metric": "x", "value": 1}\n')

    monkeypatch.setattr(bench.subprocess, "run", fake_run)
    for var in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        monkeypatch.delenv(var, raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2", "--steps", "3", "--workload", "cfg4"])
    with pytest.raises(SystemExit) as ei:
        bench.main()
    assert ei.value.code == 7
    captured = capsys.readouterr()
    assert captured.out == '{"metric": "x", "value": 1}\n' and "[Gloo]" in captured.err      # stdout = the one JSON line
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert cmd[cmd.index("--nproc-per-node") + 1] == "2" and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert 1024 < int(cmd[cmd.index("--master-port") + 1]) < 65536
    assert cmd[-6:] == ["--gpus", "2", "--steps", "3", "--workload", "cfg4"] and cmd[-7].endswith("bench.py")
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    assert not torch.cuda.is_initialized()
    # under a launcher (RANK set) the same command line is a rank, not a launcher: it must not spawn again
    monkeypatch.setenv("RANK", "0")
    monkeypatch.setenv("WORLD_SIZE", "3")
    seen.clear()
    with pytest.raises(SystemExit, match="WORLD_SIZE=3"):
        bench.main()
    assert not seen


def test_bench_rank_body_runs_world_2_over_gloo_without_a_gpu(tmp_path):
    """bench.py's N > 1 branch had only ever executed on the GPU box's one-GPU rehearsal.  Here the unmodified
    bench.main() runs as two rank processes over gloo with the engine replaced by the C oracle (tests/bench_cpu_rank.py): the
    barrier-bracketed region, the MAX over ranks, per-rank seeds and gates, parity.all_ranks_ok, gather_ms and ranks_seen."""
    import json
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    argv = ["--gpus", "2", "--batch", "384", "--steps", "2", "--warmup", "1", "--no-configs", "--no-pcie", "--no-alongside", "--no-ct", "--full-json", str(tmp_path / "full.json")]
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="2")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "bench_cpu_rank.py")] + argv, env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=600) for p in procs]
    assert all(p.returncode == 0 for p in procs), "\\n".join(o[1][-2000:] for o in outs)
    lines0 = [ln for ln in outs[0][0].splitlines() if ln.startswith('{"metric"')]
    assert len(lines0) == 1 and not [ln for ln in outs[1][0].splitlines() if ln.startswith('{"metric"')]      # rank 0 alone speaks
    assert len(lines0[0]) <= 6144 and [ln for ln in outs[0][0].splitlines() if ln.strip()] == lines0      # the driver's reader: one short line, nothing else
    line = json.loads(lines0[0])
    full = json.loads([ln for ln in outs[0][1].splitlines() if ln.startswith('{"metric"')][-1])                # the full record: rank 0's stderr
    assert line["n_gpus"] == 2 and line["config"]["ranks_seen"] == 2 and line["scaling"] == "weak"
    assert line["config"]["backend"].startswith("gloo")
    assert line["parity_ok"] is True and full["parity"]["ok"] is True and full["parity"]["all_ranks_ok"] is True and full["parity"]["units"] == 384
    assert full["parity"]["edge_cases_checked"] == 54                                                           # the edge-case mini-batch ran on the ranks too
    assert line["gather_ms"] > 0 and line["steps"] == 2 and "cpu_baseline" not in line
    lo, hi = line["cycles_per_unit_ranks"]
    assert 0 < lo <= hi
    # whole-job value: both ranks' units over the slower rank's time
    assert abs(line["value"] - 2 * 384 * 2 / (line["ms_per_step"] * 2 * 1e-3)) / line["value"] < 0.01
