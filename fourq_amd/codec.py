"""Host-side conversion between the reference's value conventions (nested tuples of Python ints,
curve4q.py) and the C ABI's little-endian 64-bit word arrays (include/fourq_amd.h)."""
import numpy as np

from .constants import P127

M64 = (1 << 64) - 1


def _fp_words(x):
    x = int(x) % P127          # the reference reduces with `% p1271` everywhere (fields.py:29-57)
    return (x & M64, x >> 64)


def pack_fp2s(elems):
    """Sequence of GF(p^2) pairs -> flat uint64 array (4 words each)."""
    out = np.empty(4 * len(elems), dtype=np.uint64)
    k = 0
    for re, im in elems:
        out[k], out[k + 1] = _fp_words(re)
        out[k + 2], out[k + 3] = _fp_words(im)
        k += 4
    return out


def pack_point(P):
    """A tuple of GF(p^2) pairs (affine: 2, R4: 3, R2/R3: 4, R1: 5) -> uint64 array."""
    return pack_fp2s(list(P))


def pack_points(points, arity):
    """List of points with `arity` coordinates each -> (n, 4*arity) uint64 array."""
    arr = np.empty((len(points), 4 * arity), dtype=np.uint64)
    for i, P in enumerate(points):
        if len(P) != arity:
            raise ValueError("expected a point with %d coordinates, got %d" % (arity, len(P)))
        arr[i] = pack_fp2s(list(P))
    return arr


def unpack_fp2s(words):
    """Flat uint64 array -> tuple of GF(p^2) pairs."""
    w = [int(x) for x in np.asarray(words, dtype=np.uint64).ravel()]
    return tuple((w[i] | (w[i + 1] << 64), w[i + 2] | (w[i + 3] << 64)) for i in range(0, len(w), 4))


def unpack_points(arr):
    return [unpack_fp2s(row) for row in np.asarray(arr, dtype=np.uint64)]


def pack_scalars(scalars):
    """Non-negative ints < 2^256 -> (n, 4) uint64 array (little-endian words, curve4q.py:552-559)."""
    arr = np.empty((len(scalars), 4), dtype=np.uint64)
    for i, m in enumerate(scalars):
        m = int(m)
        if m < 0 or m >> 256:
            raise ValueError("scalar out of range [0, 2^256)")
        arr[i] = [(m >> (64 * k)) & M64 for k in range(4)]
    return arr


def unpack_scalars(arr):
    return [sum(int(w) << (64 * k) for k, w in enumerate(row)) for row in np.asarray(arr, dtype=np.uint64)]


def pack_table(T):
    """List of 8 R2 tuples -> (128,) uint64 array."""
    if len(T) != 8:
        raise ValueError("a table has 8 entries")
    return pack_points(list(T), 4).ravel()


def unpack_table(words):
    return [unpack_fp2s(row) for row in np.asarray(words, dtype=np.uint64).reshape(8, 16)]


def scalars_from_bytes(raw):
    """n*32 random bytes -> (n, 4) uint64 scalars (uniform on [0, 2^256))."""
    return np.frombuffer(raw, dtype="<u8").reshape(-1, 4).copy()
