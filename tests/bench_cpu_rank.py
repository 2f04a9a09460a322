"""TEST INFRASTRUCTURE.  One rank of `bench.py --gpus N` on a box WITHOUT a GPU: bench.GPU (the object through which the rank body
reaches torch.cuda and the engine) is replaced by a CPU stand-in whose "engine" is the C oracle, and the unmodified bench.main()
runs under the environment a launcher would give it (RANK, WORLD_SIZE, MASTER_*; FOURQ_BENCH_REHEARSE=1 selects gloo).  What this
exercises is everything of the N > 1 path that is not a kernel: the barrier-bracketed timed region, the MAX over ranks, the per-rank
shard seeds and parity gates, `parity.all_ranks_ok` (a MIN over ranks), `gather_ms` (fourq_amd.dist.gather_rows), `ranks_seen`,
and that rank 0 alone prints one JSON line.  The numbers it prints are oracle timings and mean nothing.  (tests/test_dist.py)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np       # noqa: E402
import torch             # noqa: E402

import bench             # noqa: E402
import oracle_c as oc    # noqa: E402


def _np(t):
    return t.numpy().view(np.uint64)


class OracleEngine:
    """The entry points bench.py's headline workload calls, answered by oracle/fourq_oracle.c on host tensors."""
    version, build_id, ct_select, lanes = 0, "cpu-stand-in", False, 65536

    def table_endo(self, p_r1):
        return oc.table(oc.ENDO, np.asarray(p_r1, dtype=np.uint64))

    def mul_endo_fixed_dev(self, scalars, table, out, n):
        _np(out)[:] = oc.mul(oc.ENDO, _np(scalars), None, table)

    def mul_endo_dev(self, scalars, points, out, n):
        _np(out)[:] = oc.mul(oc.ENDO, _np(scalars), _np(points))
        if os.environ.get("FOURQ_STANDIN_CORRUPT_RANK") == os.environ.get("RANK"):       # tests: this rank's gate must catch it and stop the job
            _np(out)[n // 2, 3] ^= 1

    # the host-array calls of the edge-case mini-batch (bench.edge_case_check: every rank, every run)
    def mul_endo(self, scalars, points_r1):
        return oc.mul(oc.ENDO, np.ascontiguousarray(scalars), np.ascontiguousarray(points_r1))

    def mul_windowed(self, scalars, points_r1):
        return oc.mul(oc.WINDOWED, np.ascontiguousarray(scalars), np.ascontiguousarray(points_r1))

    def dh_endo(self, scalars, points_affine):
        return oc.dh(oc.ENDO, np.ascontiguousarray(scalars), np.ascontiguousarray(points_affine))

    def diag_clock(self, window_us):
        return {"mhz": 1000.0, "mhz_min": 1000.0, "mhz_max": 1000.0, "window_us": window_us}     # no shader clock on a CPU: a placeholder

    def diag_clock_begin(self):
        self._t0, self._t1 = time.perf_counter(), None

    def diag_clock_stop(self):
        self._t1 = time.perf_counter()

    def diag_clock_end(self):
        self._t1 = getattr(self, "_t1", None) or time.perf_counter()
        return {"mhz": 1000.0, "mhz_min": 1000.0, "mhz_max": 1000.0, "window_us": (self._t1 - self._t0) * 1e6}

    def close(self):
        pass


class _Event:
    def record(self, stream=None):
        self.t = time.perf_counter()

    def elapsed_time(self, other):
        return (other.t - self.t) * 1e3


class CpuStandIn:
    def set_device(self, index):
        pass

    def device(self, index):
        return torch.device("cpu")

    def new_stream(self, device):
        return None

    def synchronize(self):
        pass

    def event(self):
        return _Event()

    def empty_cache(self):
        pass

    def engine(self, index, stream):
        return OracleEngine()


if __name__ == "__main__":
    bench.GPU = CpuStandIn()
    os.environ["FOURQ_BENCH_REHEARSE"] = "1"
    os.environ["FOURQ_BENCH_SETTLE_MS"] = "0"
    oc.lib().fqo_set_num_threads(2)
    bench.main()
