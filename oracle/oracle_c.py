"""ctypes access to the C oracle (oracle/fourq_oracle.c) for the tests and for the cpu_baseline leg of bench.py.
TEST INFRASTRUCTURE: nothing under fourq_amd/ may import this."""
import ctypes
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
SO = os.path.join(ORACLE_DIR, "_build", "libfourq_oracle.so")


def build():
    src = os.path.join(ORACLE_DIR, "fourq_oracle.c")
    if not os.path.exists(SO) or os.path.getmtime(SO) < os.path.getmtime(src):
        subprocess.run(["make", "-C", ORACLE_DIR], check=True, capture_output=True)
    return SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(build())
        vp, sz, i = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int
        _lib.fqo_table_windowed.argtypes = [vp, vp]
        _lib.fqo_table_endo.argtypes = [vp, vp]
        _lib.fqo_mul_batch.argtypes = [i, vp, vp, vp, vp, sz]
        _lib.fqo_dh_batch.argtypes = [i, vp, vp, vp, vp, vp, sz]
        _lib.fqo_decompose_batch.argtypes = [vp, vp, sz]
        _lib.fqo_r1_to_affine_batch.argtypes = [vp, vp, sz]
        _lib.fqo_r1_to_affine_batch.restype = None
        _lib.fqo_num_threads.restype = ctypes.c_int
        _lib.fqo_set_num_threads.argtypes = [i]
        _lib.fqo_set_num_threads.restype = None
        for f in (_lib.fqo_table_windowed, _lib.fqo_table_endo, _lib.fqo_mul_batch, _lib.fqo_dh_batch, _lib.fqo_decompose_batch):
            f.restype = None
    return _lib


def _p(a):
    return None if a is None else ctypes.c_void_p(a.ctypes.data)


def _u64(a, cols):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    return a.reshape(-1, cols) if cols else a.ravel()


ENDO, WINDOWED = 0, 1


def table(kind, p_r1):
    p = _u64(p_r1, None)
    out = np.empty(128, dtype=np.uint64)
    (lib().fqo_table_endo if kind == ENDO else lib().fqo_table_windowed)(_p(p), _p(out))
    return out


def mul(kind, scalars, points_r1=None, table_words=None):
    s = _u64(scalars, 4)
    pts = None if points_r1 is None else _u64(points_r1, 20)
    tb = None if table_words is None else _u64(table_words, None)
    out = np.empty((len(s), 20), dtype=np.uint64)
    lib().fqo_mul_batch(kind, _p(s), _p(pts), _p(tb), _p(out), len(s))
    return out


def dh(kind, scalars, points_affine, table_words=None):
    s, pts = _u64(scalars, 4), _u64(points_affine, 8)
    tb = None if table_words is None else _u64(table_words, None)
    out = np.empty((len(s), 8), dtype=np.uint64)
    status = np.empty(len(s), dtype=np.uint8)
    lib().fqo_dh_batch(kind, _p(s), _p(pts), _p(tb), _p(out), _p(status), len(s))
    return out, status


def r1_to_affine(points_r1):
    """R1toAffine (curve4q.py:103-106) of every row -> (n, 8) canonical affine words."""
    pts = _u64(points_r1, 20)
    out = np.empty((len(pts), 8), dtype=np.uint64)
    lib().fqo_r1_to_affine_batch(_p(pts), _p(out), len(pts))
    return out


def encode(points_affine):
    """encode (curve4q.py:41-46) of canonical affine rows -> (n, 32) uint8: y little-endian, top bit = sign(x) (:33-39)."""
    a = _u64(points_affine, 8)
    words = a[:, 4:8].copy()
    x0_zero = (a[:, 0] | a[:, 1]) == 0
    sign = np.where(x0_zero, a[:, 3] >> np.uint64(62), a[:, 1] >> np.uint64(62)) & np.uint64(1)
    words[:, 3] |= sign << np.uint64(63)
    return words.view(np.uint8).reshape(-1, 32)


def decompose(scalars):
    s = _u64(scalars, 4)
    out = np.empty_like(s)
    lib().fqo_decompose_batch(_p(s), _p(out), len(s))
    return out


def num_threads():
    return int(lib().fqo_num_threads())


def set_num_threads(n):
    """OpenMP threads of the batch entry points (OMP_NUM_THREADS is only read when libgomp loads)."""
    lib().fqo_set_num_threads(int(n))
    return num_threads()
