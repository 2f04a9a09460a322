"""Batched FourQ engine: a thin object over one fourq_ctx (one GPU, one HIP stream).

Array conventions are the C ABI's (include/fourq_amd.h): uint64 little-endian words,
scalars (n,4), affine points (n,8), R1 points (n,20), tables (128,).  Host entry points take and
return numpy arrays; the `*_dev` entry points take device pointers (ints, or anything with a
`data_ptr()` such as a torch tensor), enqueue on the engine's stream and do not synchronise.
"""
import ctypes
import os
import threading

import numpy as np

from . import _lib
from ._lib import PRIM, FourQError, HostStats, check


def _ptr(x):
    if x is None:
        return None
    if isinstance(x, int):
        return ctypes.c_void_p(x)
    if isinstance(x, np.ndarray):
        return ctypes.c_void_p(x.ctypes.data)
    if hasattr(x, "data_ptr"):
        return ctypes.c_void_p(x.data_ptr())
    raise TypeError("expected a pointer, numpy array or tensor, got %r" % type(x))


def _host(a, cols, dtype=np.uint64):
    a = np.ascontiguousarray(a, dtype=dtype)
    if cols is not None:
        a = a.reshape(-1, cols)
    return a


def _out(out, n, cols, dtype=np.uint64):
    """A caller-provided result array (e.g. a row slice of a bigger one, or pinned memory from host_empty) or a fresh one."""
    shape = (n,) if cols is None else (n, cols)
    if out is None:
        return np.empty(shape, dtype=dtype)
    if out.dtype != dtype or out.shape != shape or not out.flags.c_contiguous:
        raise ValueError("out must be a C-contiguous %s %s array" % (shape, np.dtype(dtype).name))
    return out


class Engine:
    def __init__(self, device=None, stream=None):
        self._lib = _lib.load()
        if device is None:
            device = int(os.environ.get("LOCAL_RANK", "0"))
        ctx = ctypes.c_void_p()
        check(self._lib.fourq_ctx_create(int(device), ctypes.byref(ctx)))
        self._ctx = ctx
        self.device = int(device)
        if stream is not None:
            self.set_stream(stream)

        self._pinned = {}            # address -> size of the pinned host blocks handed out by host_empty()

    # ---- lifetime --------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_ctx", None):
            for addr in list(getattr(self, "_pinned", {})):
                self._lib.fourq_host_free(self._ctx, ctypes.c_void_p(addr))
            self._pinned = {}
            self._lib.fourq_ctx_destroy(self._ctx)
            self._ctx = None

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _ck(self, rc):
        check(rc, self._ctx)

    def set_stream(self, stream):
        """`stream`: a hipStream_t as int (e.g. torch.cuda.current_stream().cuda_stream) or None."""
        handle = getattr(stream, "cuda_stream", stream)
        self._ck(self._lib.fourq_ctx_set_stream(self._ctx, ctypes.c_void_p(handle) if handle else None))

    def sync(self):
        self._ck(self._lib.fourq_ctx_sync(self._ctx))

    @property
    def ct_select(self):
        """Constant-time table selection (fourq_ctx_set_ct_select): False by default -- the digit is an address, as in
        the reference; True reads the whole table at every step.  Results are identical."""
        on = ctypes.c_int()
        self._ck(self._lib.fourq_ctx_get_ct_select(self._ctx, ctypes.byref(on)))
        return bool(on.value)

    @ct_select.setter
    def ct_select(self, on):
        self._ck(self._lib.fourq_ctx_set_ct_select(self._ctx, 1 if on else 0))

    # ---- pinned host memory (the fast path of the host-pointer calls) ---------------------------
    def host_empty(self, shape, dtype=np.uint64):
        """Uninitialised numpy array in pinned host memory (fourq_host_alloc): the host-array entry points move such
        arrays by DMA without a bounce copy.  The memory belongs to the engine and is released by close() (or
        host_free); arrays must not be used after that."""
        dt = np.dtype(dtype)
        count = int(np.prod(shape, dtype=np.int64)) if np.ndim(shape) else int(shape)
        nbytes = max(16, count * dt.itemsize)
        ptr = ctypes.c_void_p()
        self._ck(self._lib.fourq_host_alloc(self._ctx, nbytes, ctypes.byref(ptr)))
        self._pinned[ptr.value] = nbytes
        buf = (ctypes.c_char * nbytes).from_address(ptr.value)
        return np.frombuffer(buf, dtype=dt, count=count).reshape(shape)

    def host_array(self, a, dtype=None):
        """Copy of `a` in pinned host memory."""
        a = np.asarray(a, dtype=dtype)
        out = self.host_empty(a.shape, a.dtype)
        out[...] = a
        return out

    def host_free(self, a):
        addr = a.ctypes.data
        if addr in self._pinned:
            self._ck(self._lib.fourq_host_free(self._ctx, ctypes.c_void_p(addr)))
            del self._pinned[addr]

    def host_timing(self, on):
        """Time the chunk copies of the following host-array calls (fourq_ctx_set_host_timing): `host_stats()` then reports h2d_ms /
        d2h_ms and the GB/s they imply.  Off by default -- the event records are not free."""
        self._ck(self._lib.fourq_ctx_set_host_timing(self._ctx, 1 if on else 0))

    def diag_clock(self, window_us=20000):
        """Shader clock in MHz the device holds right now, measured inside a kernel over `window_us` (fourq_diag_clock): dict with median,
        min, max over the probe's 16 waves and `under_load` (the engine's stream still had work in flight when the window closed).  Call
        it with work queued on the engine's stream for longer than the window."""
        med, lo, hi, busy = ctypes.c_double(), ctypes.c_double(), ctypes.c_double(), ctypes.c_int()
        self._ck(self._lib.fourq_diag_clock(self._ctx, int(window_us), ctypes.byref(med), ctypes.byref(lo), ctypes.byref(hi), ctypes.byref(busy)))
        return {"mhz": med.value, "mhz_min": lo.value, "mhz_max": hi.value, "window_us": int(window_us), "under_load": bool(busy.value)}

    def diag_clock_begin(self):
        """First stamp of a clock bracket, enqueued on the engine's stream (fourq_diag_clock_begin): the clock reported by diag_clock_end()
        is that of everything enqueued between this and diag_clock_stop()."""
        self._ck(self._lib.fourq_diag_clock_begin(self._ctx))

    def diag_clock_stop(self):
        """Second stamp, enqueued behind the bracketed work (fourq_diag_clock_stop); the host does not wait."""
        self._ck(self._lib.fourq_diag_clock_stop(self._ctx))

    def diag_clock_end(self):
        med, lo, hi, win = ctypes.c_double(), ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
        self._ck(self._lib.fourq_diag_clock_end(self._ctx, ctypes.byref(med), ctypes.byref(lo), ctypes.byref(hi), ctypes.byref(win)))
        return {"mhz": med.value, "mhz_min": lo.value, "mhz_max": hi.value, "window_us": win.value}

    def host_stats(self):
        """Transfer statistics of the last host-array call: bytes, chunks, pinned flags; copy milliseconds and GB/s under host_timing(True)."""
        st = HostStats()
        self._ck(self._lib.fourq_ctx_host_stats(self._ctx, ctypes.byref(st)))
        d = {f: getattr(st, f) for f, _ in HostStats._fields_}
        d["gbs_h2d"] = st.h2d_bytes / st.h2d_ms / 1e6 if st.h2d_ms > 0 else None
        d["gbs_d2h"] = st.d2h_bytes / st.d2h_ms / 1e6 if st.d2h_ms > 0 else None
        return d

    def host_chunk_stamps(self):
        """Per chunk of the last host-array call made under host_timing(True): [copy-in start, copy-in end, copy-out start, copy-out end,
        kernels start, kernels end] in ms since the call's first event (fourq_ctx_host_chunk_stamps)."""
        rows, buf = [], (ctypes.c_double * 6)()
        while self._lib.fourq_ctx_host_chunk_stamps(self._ctx, len(rows), buf) == 0:
            rows.append(list(buf))
        return rows

    def reserve(self, n):
        """Size the context's internal buffers for `*_dev` batches of up to n elements (fourq_ctx_reserve): such calls then
        only enqueue -- no hidden stream synchronisation, capturable into a HIP graph."""
        self._ck(self._lib.fourq_ctx_reserve(self._ctx, int(n)))

    @property
    def build_id(self):
        return self._lib.fourq_build_id().decode()

    @property
    def version(self):
        return int(self._lib.fourq_version())

    @property
    def lanes(self):
        n = ctypes.c_size_t()
        self._ck(self._lib.fourq_ctx_lanes(self._ctx, ctypes.byref(n)))
        return n.value

    # ---- tables ----------------------------------------------------------------------------
    def _table(self, fn, p_r1):
        p = _host(p_r1, None).ravel()
        if p.size != 20:
            raise ValueError("an R1 point is 20 words")
        out = np.empty(128, dtype=np.uint64)
        self._ck(fn(self._ctx, _ptr(p), _ptr(out)))
        return out

    def table_endo(self, p_r1):
        return self._table(self._lib.fourq_table_endo, p_r1)

    def table_windowed(self, p_r1):
        return self._table(self._lib.fourq_table_windowed, p_r1)

    # ---- scalar multiplication (host arrays) -----------------------------------------------
    def _mul(self, fn, scalars, second, second_cols, out=None):
        s = _host(scalars, 4)
        b = _host(second, second_cols)
        if second_cols == 20 and len(b) != len(s):
            raise ValueError("scalars and points differ in length")
        if second_cols is None and b.size != 128:
            raise ValueError("a table is 128 words")
        out = _out(out, len(s), 20)
        self._ck(fn(self._ctx, _ptr(s), _ptr(b), _ptr(out), len(s)))
        return out

    # `out`: optional preallocated result array, e.g. from host_empty() (pinned: no bounce copy on the way back)
    def mul_endo(self, scalars, points_r1, out=None):
        return self._mul(self._lib.fourq_mul_endo_batch, scalars, points_r1, 20, out)

    def mul_windowed(self, scalars, points_r1, out=None):
        return self._mul(self._lib.fourq_mul_windowed_batch, scalars, points_r1, 20, out)

    def mul_affine(self, scalars, points_affine, kind="endo", out=None):
        """R1toAffine(MUL_<kind>(m_i, AffineToR1(P_i))) (curve4q.py:100-106): affine in, canonical affine out -- 160 bytes per
        operation across the link instead of the raw-R1 form's 352 (fourq_mul_*_affine_batch)."""
        s, p = _host(scalars, 4), _host(points_affine, 8)
        if len(s) != len(p):
            raise ValueError("scalars and points differ in length")
        out = _out(out, len(s), 8)
        fn = self._lib.fourq_mul_endo_affine_batch if kind == "endo" else self._lib.fourq_mul_windowed_affine_batch
        self._ck(fn(self._ctx, _ptr(s), _ptr(p), _ptr(out), len(s)))
        return out

    def mul_bytes(self, scalars, points32, kind="endo", out=None, status=None):
        """encode(R1toAffine(MUL_<kind>(m_i, AffineToR1(decode(B_i))))) (curve4q.py:41-96): 32-byte points in and out, 96 bytes per
        operation (fourq_mul_*_bytes_batch).  Returns ((n, 32) uint8, status): 0 ok, 16 + decode status."""
        s, b = _host(scalars, 4), _host(points32, 32, np.uint8)
        if len(s) != len(b):
            raise ValueError("scalars and points differ in length")
        out, status = _out(out, len(s), 32, np.uint8), _out(status, len(s), None, np.uint8)
        fn = self._lib.fourq_mul_endo_bytes_batch if kind == "endo" else self._lib.fourq_mul_windowed_bytes_batch
        self._ck(fn(self._ctx, _ptr(s), _ptr(b), _ptr(out), _ptr(status), len(s)))
        return out, status

    def mul_affine_dev(self, scalars, points_affine, out_affine, n, kind="endo"):
        fn = self._lib.fourq_mul_endo_affine_batch_dev if kind == "endo" else self._lib.fourq_mul_windowed_affine_batch_dev
        self._ck(fn(self._ctx, _ptr(scalars), _ptr(points_affine), _ptr(out_affine), n))

    def mul_bytes_dev(self, scalars, points32, out32, status, n, kind="endo"):
        fn = self._lib.fourq_mul_endo_bytes_batch_dev if kind == "endo" else self._lib.fourq_mul_windowed_bytes_batch_dev
        self._ck(fn(self._ctx, _ptr(scalars), _ptr(points32), _ptr(out32), _ptr(status), n))

    def mul_endo_fixed(self, scalars, table, out=None):
        return self._mul(self._lib.fourq_mul_endo_fixed_batch, scalars, table, None, out)

    def mul_windowed_fixed(self, scalars, table, out=None):
        return self._mul(self._lib.fourq_mul_windowed_fixed_batch, scalars, table, None, out)

    def mul_endo_mixed(self, scalars, points_r1, flags, table, out=None):
        s, p = _host(scalars, 4), _host(points_r1, 20)
        f = _host(flags, None, np.uint8).ravel()
        t = _host(table, None).ravel()
        if not (len(s) == len(p) == len(f)) or t.size != 128:
            raise ValueError("mixed batch: inconsistent shapes")
        out = _out(out, len(s), 20)
        self._ck(self._lib.fourq_mul_endo_mixed_batch(self._ctx, _ptr(s), _ptr(p), _ptr(f), _ptr(t), _ptr(out), len(s)))
        return out

    # ---- Diffie-Hellman ----------------------------------------------------------------------
    def _dh(self, fn, scalars, points_affine, table, out=None, status=None):
        s, p = _host(scalars, 4), _host(points_affine, 8)
        if len(s) != len(p):
            raise ValueError("scalars and points differ in length")
        t = None if table is None else _host(table, None).ravel()
        if t is not None and t.size != 128:
            raise ValueError("a table is 128 words")
        out, status = _out(out, len(s), 8), _out(status, len(s), None, np.uint8)
        self._ck(fn(self._ctx, _ptr(s), _ptr(p), _ptr(t), _ptr(out), _ptr(status), len(s)))
        return out, status

    def dh_endo(self, scalars, points_affine, table=None, out=None, status=None):
        return self._dh(self._lib.fourq_dh_endo_batch, scalars, points_affine, table, out, status)

    def dh_windowed(self, scalars, points_affine, table=None, out=None, status=None):
        return self._dh(self._lib.fourq_dh_windowed_batch, scalars, points_affine, table, out, status)

    def dh_exchange(self, a_scalars, b_scalars, base_affine, table392=None, out=None, status=None):
        """One exchange per row: DH_endo(a_i, DH_endo(b_i, base)) (curve4q.py:731), the first half's
        public keys staying on the device (fourq_dh_exchange_batch).

        `table392` = table_endo([392]base) makes the first half fixed-base.  Returns (affine, status):
        status is the first failure of either half."""
        a, b = _host(a_scalars, 4), _host(b_scalars, 4)
        if len(a) != len(b):
            raise ValueError("the two scalar arrays differ in length")
        base = _host(base_affine, None).ravel()
        if base.size != 8:
            raise ValueError("an affine point is 8 words")
        t = None if table392 is None else _host(table392, None).ravel()
        if t is not None and t.size != 128:
            raise ValueError("a table is 128 words")
        out, status = _out(out, len(a), 8), _out(status, len(a), None, np.uint8)
        self._ck(self._lib.fourq_dh_exchange_batch(self._ctx, _ptr(a), _ptr(b), _ptr(base), _ptr(t), _ptr(out), _ptr(status), len(a)))
        return out, status

    def dh_exchange_comb(self, a_scalars, b_scalars, comb=None, out=None, status=None):
        """One exchange per row with the key-generation half through the comb of [392]base (fourq_dh_exchange_comb_batch):
        DH_endo(a_i, DH_endo(b_i, base)) -- same outputs as dh_exchange, the first half 3x cheaper.  `comb`: a comb table
        (comb_table([392]base)) or None = the table staged by comb_stage().  Returns (affine, status)."""
        a, b = _host(a_scalars, 4), _host(b_scalars, 4)
        if len(a) != len(b):
            raise ValueError("the two scalar arrays differ in length")
        t = None
        if comb is not None:
            t = _host(comb, None).ravel()
            if t.size != _lib.COMB_WORDS:
                raise ValueError("a comb table is %d words" % _lib.COMB_WORDS)
        out, status = _out(out, len(a), 8), _out(status, len(a), None, np.uint8)
        self._ck(self._lib.fourq_dh_exchange_comb_batch(self._ctx, _ptr(a), _ptr(b), _ptr(t), _ptr(out), _ptr(status), len(a)))
        return out, status

    def dh_exchange_comb_dev(self, a_scalars, b_scalars, comb_host, out_affine, status, n):
        t = None if comb_host is None else _host(comb_host, None).ravel()
        self._ck(self._lib.fourq_dh_exchange_comb_batch_dev(self._ctx, _ptr(a_scalars), _ptr(b_scalars), _ptr(t), _ptr(out_affine), _ptr(status), n))

    def dh_exchange_dev(self, a_scalars, b_scalars, base_affine_host, table392_host, out_affine, status, n):
        base = _host(base_affine_host, None).ravel()
        t = None if table392_host is None else _host(table392_host, None).ravel()
        self._ck(self._lib.fourq_dh_exchange_batch_dev(self._ctx, _ptr(a_scalars), _ptr(b_scalars), _ptr(base), _ptr(t), _ptr(out_affine), _ptr(status), n))

    # ---- fixed-base comb (1024-point table; affine outputs only) -----------------------------------
    def comb_table(self, p_r1):
        """Comb table (_lib.COMB_WORDS words) of the order-N point `p_r1` (fourq_comb_table)."""
        p = _host(p_r1, None).ravel()
        if p.size != 20:
            raise ValueError("an R1 point is 20 words")
        out = np.empty(_lib.COMB_WORDS, dtype=np.uint64)
        self._ck(self._lib.fourq_comb_table(self._ctx, _ptr(p), _ptr(out)))
        return out

    def comb_mul(self, scalars, comb, out=None, status=None):
        """Affine [m_i]B for the comb's base B: ((n, 8) words, status) -- equals R1toAffine(MUL_endo(m_i, B))."""
        s = _host(scalars, 4)
        t = _host(comb, None).ravel()
        if t.size != _lib.COMB_WORDS:
            raise ValueError("a comb table is %d words" % _lib.COMB_WORDS)
        out, status = _out(out, len(s), 8), _out(status, len(s), None, np.uint8)
        self._ck(self._lib.fourq_comb_mul_batch(self._ctx, _ptr(s), _ptr(t), _ptr(out), _ptr(status), len(s)))
        return out, status

    def comb_stage(self, comb):
        """Upload a comb table once (fourq_comb_stage); comb_mul_dev(..., comb_host=None, ...) then uses it with no per-call
        work on the table."""
        t = _host(comb, None).ravel()
        if t.size != _lib.COMB_WORDS:
            raise ValueError("a comb table is %d words" % _lib.COMB_WORDS)
        self._ck(self._lib.fourq_comb_stage(self._ctx, _ptr(t)))

    def comb_mul_dev(self, scalars, comb_host, out_affine, status, n):
        """`comb_host`: the table (compared with the staged copy on every call) or None = the table staged by comb_stage()."""
        t = None
        if comb_host is not None:
            t = _host(comb_host, None).ravel()
            if t.size != _lib.COMB_WORDS:
                raise ValueError("a comb table is %d words" % _lib.COMB_WORDS)
        self._ck(self._lib.fourq_comb_mul_batch_dev(self._ctx, _ptr(scalars), _ptr(t), _ptr(out_affine), _ptr(status), n))

    # ---- point compression (32-byte wire format) -------------------------------------------------
    def encode(self, points_affine, out=None):
        """(n, 8) affine words -> (n, 32) uint8 encodings (curve4q.py:41-46)."""
        p = _host(points_affine, 8)
        out = _out(out, len(p), 32, np.uint8)
        self._ck(self._lib.fourq_encode_batch(self._ctx, _ptr(p), _ptr(out), len(p)))
        return out

    def decode(self, encodings, out=None, status=None):
        """(n, 32) uint8 -> ((n, 8) affine words, (n,) status) (curve4q.py:49-96; status = _lib.DECODE_*)."""
        b = _host(encodings, 32, np.uint8)
        out, status = _out(out, len(b), 8), _out(status, len(b), None, np.uint8)
        self._ck(self._lib.fourq_decode_batch(self._ctx, _ptr(b), _ptr(out), _ptr(status), len(b)))
        return out, status

    def dh_bytes(self, scalars, public_keys32, kind="endo", table=None, out=None, status=None):
        """The protocol step of draft-ladd-cfrg-4q section "Diffie-Hellman": decode each 32-byte public key, DH_<kind> with
        the scalar, encode the shared point -- one call, intermediates stay on the GPU (fourq_dh_*_bytes_batch).
        Returns ((n, 32) uint8, status): 0 ok, 1/2 as DH_*, 16 + decode status."""
        s, k = _host(scalars, 4), _host(public_keys32, 32, np.uint8)
        if len(s) != len(k):
            raise ValueError("scalars and keys differ in length")
        t = None if table is None else _host(table, None).ravel()
        if t is not None and t.size != 128:
            raise ValueError("a table is 128 words")
        out, status = _out(out, len(s), 32, np.uint8), _out(status, len(s), None, np.uint8)
        fn = self._lib.fourq_dh_endo_bytes_batch if kind == "endo" else self._lib.fourq_dh_windowed_bytes_batch
        self._ck(fn(self._ctx, _ptr(s), _ptr(k), _ptr(t), _ptr(out), _ptr(status), len(s)))
        return out, status

    def dh_bytes_dev(self, scalars, keys32, table_host, out32, status, n, kind="endo"):
        t = None if table_host is None else _host(table_host, None).ravel()
        fn = self._lib.fourq_dh_endo_bytes_batch_dev if kind == "endo" else self._lib.fourq_dh_windowed_bytes_batch_dev
        self._ck(fn(self._ctx, _ptr(scalars), _ptr(keys32), _ptr(t), _ptr(out32), _ptr(status), n))

    def encode_dev(self, points_affine, out32, n):
        self._ck(self._lib.fourq_encode_batch_dev(self._ctx, _ptr(points_affine), _ptr(out32), n))

    def decode_dev(self, in32, out_affine, status, n):
        self._ck(self._lib.fourq_decode_batch_dev(self._ctx, _ptr(in32), _ptr(out_affine), _ptr(status), n))

    # ---- device-pointer flavour (async on the engine's stream) ---------------------------------
    def mul_endo_dev(self, scalars, points_r1, out_r1, n):
        self._ck(self._lib.fourq_mul_endo_batch_dev(self._ctx, _ptr(scalars), _ptr(points_r1), _ptr(out_r1), n))

    def mul_windowed_dev(self, scalars, points_r1, out_r1, n):
        self._ck(self._lib.fourq_mul_windowed_batch_dev(self._ctx, _ptr(scalars), _ptr(points_r1), _ptr(out_r1), n))

    def mul_endo_fixed_dev(self, scalars, table_host, out_r1, n):
        t = _host(table_host, None).ravel()
        self._ck(self._lib.fourq_mul_endo_fixed_batch_dev(self._ctx, _ptr(scalars), _ptr(t), _ptr(out_r1), n))

    def mul_windowed_fixed_dev(self, scalars, table_host, out_r1, n):
        t = _host(table_host, None).ravel()
        self._ck(self._lib.fourq_mul_windowed_fixed_batch_dev(self._ctx, _ptr(scalars), _ptr(t), _ptr(out_r1), n))

    def mul_endo_mixed_dev(self, scalars, points_r1, flags, table_host, out_r1, n):
        t = _host(table_host, None).ravel()
        self._ck(self._lib.fourq_mul_endo_mixed_batch_dev(self._ctx, _ptr(scalars), _ptr(points_r1), _ptr(flags), _ptr(t), _ptr(out_r1), n))

    def dh_endo_dev(self, scalars, points_affine, table_host, out_affine, status, n):
        t = None if table_host is None else _host(table_host, None).ravel()
        self._ck(self._lib.fourq_dh_endo_batch_dev(self._ctx, _ptr(scalars), _ptr(points_affine), _ptr(t), _ptr(out_affine), _ptr(status), n))

    def dh_windowed_dev(self, scalars, points_affine, table_host, out_affine, status, n):
        t = None if table_host is None else _host(table_host, None).ravel()
        self._ck(self._lib.fourq_dh_windowed_batch_dev(self._ctx, _ptr(scalars), _ptr(points_affine), _ptr(t), _ptr(out_affine), _ptr(status), n))

    # ---- primitives ----------------------------------------------------------------------------
    def prim(self, op, inputs):
        """Run primitive `op` (name in _lib.PRIM or its int) over rows of `inputs`; returns (n, out_words)."""
        code = PRIM[op] if isinstance(op, str) else int(op)
        iw, ow = ctypes.c_size_t(), ctypes.c_size_t()
        self._ck(self._lib.fourq_prim_words(code, ctypes.byref(iw), ctypes.byref(ow)))
        x = _host(inputs, iw.value)
        out = np.zeros((len(x), ow.value), dtype=np.uint64)
        self._ck(self._lib.fourq_prim_batch(self._ctx, code, _ptr(x), _ptr(out), len(x)))
        return out


_default = None
_default_lock = threading.Lock()


def default_engine():
    """Process-wide engine on device LOCAL_RANK (or 0).  Raises FourQError when no MI355X is usable."""
    global _default
    if _default is None:
        with _default_lock:                    # first calls from several threads at once: one context, not one each
            if _default is None:
                _default = Engine()
    return _default


__all__ = ["Engine", "default_engine", "FourQError"]
