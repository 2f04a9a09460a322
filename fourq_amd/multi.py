"""All GPUs of a node behind ONE call: `MultiEngine` shards the rows of a host-array batch contiguously over one
`Engine` (one fourq_ctx, one HIP stream) per device and runs the shards side by side, one host thread per device.

SURVEY.md section 8(e): every (scalar, point) pair is independent, so there is no exchange step and no collective --
shard g owns rows [g*n/G, (g+1)*n/G) (`dist.shard_bounds`, the same cut the one-process-per-GPU path of bench.py and
`dist.sharded_map` use) and writes its results straight into its rows of the caller's output arrays.  ctypes drops the
GIL for the duration of a library call, so the per-device pipelines (H2D, kernels, D2H) really
overlap.  The reference's API is a plain function call (curve4q.py:188, :405, :464-468); this is the same call with
the node's GPUs behind it.  Fixed-base tables and comb tables are replicated (1 KiB / 13 KiB per device).

    with MultiEngine() as eng:                 # every visible MI355X; MultiEngine([0, 0]) = two contexts on GPU 0
        out = eng.mul_endo(scalars, points)    # same arrays, same results as Engine.mul_endo
"""
import ctypes
from concurrent.futures import ThreadPoolExecutor

import numpy as np

from . import _lib
from .dist import shard_bounds
from .engine import Engine, _host, _out


def device_count():
    """Usable gfx950 devices as the library sees them (fourq_device_count); 0 without a GPU."""
    n = ctypes.c_int()
    _lib.check(_lib.load().fourq_device_count(ctypes.byref(n)))
    return n.value


class MultiEngine:
    def __init__(self, devices=None):
        if devices is None:
            devices = list(range(device_count()))
        devices = [int(d) for d in devices]
        if not devices:
            raise _lib.FourQError("fourq_amd: no usable gfx950 HIP device (MultiEngine needs at least one; there is no CPU fallback)")
        self.devices = devices
        self.engines = []
        try:
            for d in devices:
                self.engines.append(Engine(d))
        except Exception:
            self.close()
            raise
        self._pool = ThreadPoolExecutor(max_workers=len(devices), thread_name_prefix="fourq-dev")

    # ---- lifetime ---------------------------------------------------------------------------------------------
    def close(self):
        pool, self._pool = getattr(self, "_pool", None), None
        if pool is not None:
            pool.shutdown(wait=True)
        for e in getattr(self, "engines", []):
            e.close()
        self.engines = []

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    @property
    def ct_select(self):
        return self.engines[0].ct_select

    @ct_select.setter
    def ct_select(self, on):
        for e in self.engines:
            e.ct_select = on

    @property
    def lanes(self):
        return sum(e.lanes for e in self.engines)

    def sync(self):
        for e in self.engines:
            e.sync()

    # ---- the sharding itself ----------------------------------------------------------------------------------
    def _sharded(self, call, rows, outs):
        """`call(engine, *row_shards, *out_shards)` on every device's contiguous shard of the row arrays `rows`, results
        landing in the matching rows of `outs`.  Errors of any shard are raised after all shards have finished."""
        n = len(rows[0])
        world = len(self.engines)

        def one(g):
            lo, hi = shard_bounds(n, g, world)
            if hi > lo:
                call(self.engines[g], *[a[lo:hi] for a in rows], *[o[lo:hi] for o in outs])

        if world == 1 or n < 2:
            for g in range(world):
                one(g)
        else:
            for f in [self._pool.submit(one, g) for g in range(world)]:
                f.result()
        return outs[0] if len(outs) == 1 else tuple(outs)

    @staticmethod
    def _same_len(*arrays):
        if len({len(a) for a in arrays}) != 1:
            raise ValueError("row arrays differ in length")

    # ---- tables (device 0; tables are data, identical on every device) --------------------------------------------
    def table_endo(self, p_r1):
        return self.engines[0].table_endo(p_r1)

    def table_windowed(self, p_r1):
        return self.engines[0].table_windowed(p_r1)

    def comb_table(self, p_r1):
        return self.engines[0].comb_table(p_r1)

    # ---- scalar multiplication ------------------------------------------------------------------------------------
    def mul_endo(self, scalars, points_r1, out=None):
        s, p = _host(scalars, 4), _host(points_r1, 20)
        self._same_len(s, p)
        return self._sharded(lambda e, s, p, o: e.mul_endo(s, p, out=o), [s, p], [_out(out, len(s), 20)])

    def mul_windowed(self, scalars, points_r1, out=None):
        s, p = _host(scalars, 4), _host(points_r1, 20)
        self._same_len(s, p)
        return self._sharded(lambda e, s, p, o: e.mul_windowed(s, p, out=o), [s, p], [_out(out, len(s), 20)])

    def mul_endo_fixed(self, scalars, table, out=None):
        s = _host(scalars, 4)
        return self._sharded(lambda e, s, o: e.mul_endo_fixed(s, table, out=o), [s], [_out(out, len(s), 20)])

    def mul_windowed_fixed(self, scalars, table, out=None):
        s = _host(scalars, 4)
        return self._sharded(lambda e, s, o: e.mul_windowed_fixed(s, table, out=o), [s], [_out(out, len(s), 20)])

    def mul_endo_mixed(self, scalars, points_r1, flags, table, out=None):
        s, p, f = _host(scalars, 4), _host(points_r1, 20), _host(flags, None, np.uint8).ravel()
        self._same_len(s, p, f)
        return self._sharded(lambda e, s, p, f, o: e.mul_endo_mixed(s, p, f, table, out=o), [s, p, f], [_out(out, len(s), 20)])

    # ---- Diffie-Hellman -------------------------------------------------------------------------------------------
    def dh_endo(self, scalars, points_affine, table=None, out=None, status=None):
        s, p = _host(scalars, 4), _host(points_affine, 8)
        self._same_len(s, p)
        return self._sharded(lambda e, s, p, o, st: e.dh_endo(s, p, table, out=o, status=st), [s, p],
                             [_out(out, len(s), 8), _out(status, len(s), None, np.uint8)])

    def dh_windowed(self, scalars, points_affine, table=None, out=None, status=None):
        s, p = _host(scalars, 4), _host(points_affine, 8)
        self._same_len(s, p)
        return self._sharded(lambda e, s, p, o, st: e.dh_windowed(s, p, table, out=o, status=st), [s, p],
                             [_out(out, len(s), 8), _out(status, len(s), None, np.uint8)])

    def dh_exchange(self, a_scalars, b_scalars, base_affine, table392=None, out=None, status=None):
        a, b = _host(a_scalars, 4), _host(b_scalars, 4)
        self._same_len(a, b)
        return self._sharded(lambda e, a, b, o, st: e.dh_exchange(a, b, base_affine, table392, out=o, status=st), [a, b],
                             [_out(out, len(a), 8), _out(status, len(a), None, np.uint8)])

    def dh_exchange_comb(self, a_scalars, b_scalars, comb, out=None, status=None):
        a, b = _host(a_scalars, 4), _host(b_scalars, 4)
        self._same_len(a, b)
        return self._sharded(lambda e, a, b, o, st: e.dh_exchange_comb(a, b, comb, out=o, status=st), [a, b],
                             [_out(out, len(a), 8), _out(status, len(a), None, np.uint8)])

    def comb_mul(self, scalars, comb, out=None, status=None):
        s = _host(scalars, 4)
        return self._sharded(lambda e, s, o, st: e.comb_mul(s, comb, out=o, status=st), [s],
                             [_out(out, len(s), 8), _out(status, len(s), None, np.uint8)])

    # ---- wire format ----------------------------------------------------------------------------------------------
    def encode(self, points_affine, out=None):
        p = _host(points_affine, 8)
        return self._sharded(lambda e, p, o: e.encode(p, out=o), [p], [_out(out, len(p), 32, np.uint8)])

    def decode(self, encodings, out=None, status=None):
        b = _host(encodings, 32, np.uint8)
        return self._sharded(lambda e, b, o, st: e.decode(b, out=o, status=st), [b],
                             [_out(out, len(b), 8), _out(status, len(b), None, np.uint8)])

    def dh_bytes(self, scalars, public_keys32, kind="endo", table=None, out=None, status=None):
        s, k = _host(scalars, 4), _host(public_keys32, 32, np.uint8)
        self._same_len(s, k)
        return self._sharded(lambda e, s, k, o, st: e.dh_bytes(s, k, kind, table, out=o, status=st), [s, k],
                             [_out(out, len(s), 32, np.uint8), _out(status, len(s), None, np.uint8)])


__all__ = ["MultiEngine", "device_count"]
