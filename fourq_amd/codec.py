"""Host-side conversion between the reference's value conventions (nested tuples of Python ints,
curve4q.py) and the C ABI's little-endian 64-bit word arrays (include/fourq_amd.h)."""
import gc
import itertools

import numpy as np

from .constants import P127

try:                                    # csrc/fastcodec.c, built in-tree by fourq_amd/build.py: the same conversions in C
    from . import _fastcodec as _fc
except ImportError:                     # not built (a source checkout before build()): the pure-Python path below, ~3 us per element
    _fc = None

M64 = (1 << 64) - 1
_to_bytes = int.to_bytes
_chain = itertools.chain.from_iterable


def _fp_words(x):
    """One GF(p) value -> (low, high) 64-bit words of its residue in [0, p)."""
    x = int(x) % P127          # the reference reduces with `% p1271` everywhere (fields.py:29-57)
    return (x & M64, x >> 64)


def _words_from_ints(flat, width):
    """Non-negative Python ints below 2^(8*width) -> uint64 array, little-endian words.  One C-level pass: int.to_bytes mapped over the
    values, one join, one frombuffer (round 3's per-word numpy stores cost 4.5 us per element; this is ~0.1 us per int)."""
    return np.frombuffer(bytearray(b"".join(map(_to_bytes, flat, itertools.repeat(width), itertools.repeat("little")))), dtype="<u8")


def _fp_ints(flat):
    """GF(p) values -> uint64 words (2 each), reduced as the reference's `% p1271` does (fields.py:29-57).  (Pure-Python path.)"""
    return _words_from_ints([int(x) % P127 for x in flat], 16)


def pack_fp2s(elems):
    """Sequence of GF(p^2) pairs -> flat uint64 array (4 words each)."""
    elems = list(elems)
    if _fc is not None:
        try:
            return np.frombuffer(bytearray(_fc.pack_fp([elems], len(elems))), dtype="<u8")
        except (OverflowError, TypeError):
            pass
    flat = list(_chain(elems))
    if len(flat) != 2 * len(elems):
        raise ValueError("a GF(p^2) element is a pair")
    return _fp_ints(flat)


def pack_point(P):
    """A tuple of GF(p^2) pairs (affine: 2, R4: 3, R2/R3: 4, R1: 5) -> uint64 array."""
    return pack_fp2s(P)


def pack_points(points, arity):
    """List of points with `arity` coordinates each -> (n, 4*arity) uint64 array."""
    if _fc is not None:
        try:
            raw = _fc.pack_fp(points, arity)
            return np.frombuffer(bytearray(raw), dtype="<u8").reshape(len(raw) // (32 * arity), 4 * arity)
        except (OverflowError, TypeError):      # values to reduce mod p, numpy integers, ...: the general path
            pass
    points = list(points)
    for P in points:
        if len(P) != arity:
            raise ValueError("expected a point with %d coordinates, got %d" % (arity, len(P)))
    flat = list(_chain(_chain(points)))
    if len(flat) != 2 * arity * len(points):
        raise ValueError("a GF(p^2) element is a pair")
    return _fp_ints(flat).reshape(len(points), 4 * arity)


def _fp2s_from_words(w):
    """list of uint64 words (Python ints) -> list of GF(p^2) pairs."""
    vals = [lo | (hi << 64) for lo, hi in zip(w[0::2], w[1::2])]
    return list(zip(vals[0::2], vals[1::2]))


def unpack_fp2s(words):
    """Flat uint64 array -> tuple of GF(p^2) pairs."""
    return tuple(_fp2s_from_words(np.asarray(words, dtype=np.uint64).ravel().tolist()))


def unpack_points(arr):
    arr = np.asarray(arr, dtype=np.uint64)
    if arr.ndim != 2:
        return [unpack_fp2s(row) for row in arr]
    arity = arr.shape[1] // 4
    if _fc is not None and arr.shape[1] == 4 * arity:
        # sixteen new objects per R1 point, none of them part of a cycle: without this the collector's generation scans are two
        # thirds of the call (0.93 -> 0.35 us per element)
        was_enabled = gc.isenabled()
        gc.disable()
        try:
            return _fc.unpack_fp(np.ascontiguousarray(arr).data, arity)
        finally:
            if was_enabled:
                gc.enable()
    pairs = _fp2s_from_words(arr.ravel().tolist())
    return list(zip(*[pairs[k::arity] for k in range(arity)]))


def pack_scalars(scalars):
    """Non-negative ints < 2^256 -> (n, 4) uint64 array (little-endian words, curve4q.py:552-559)."""
    if _fc is not None:
        try:
            raw = _fc.pack_scalars(scalars)
            return np.frombuffer(bytearray(raw), dtype="<u8").reshape(len(raw) // 32, 4)
        except OverflowError:
            raise ValueError("scalar out of range [0, 2^256)") from None
        except TypeError:
            pass
    scalars = list(scalars)
    try:
        return _words_from_ints(scalars, 32).reshape(len(scalars), 4)
    except OverflowError:
        raise ValueError("scalar out of range [0, 2^256)") from None
    except TypeError:
        return pack_scalars([int(m) for m in scalars])


def unpack_scalars(arr):
    w = np.ascontiguousarray(np.asarray(arr, dtype=np.uint64).reshape(-1, 4))
    if _fc is not None:
        return _fc.unpack_scalars(w.data)
    b = w.tobytes()
    return [int.from_bytes(b[i:i + 32], "little") for i in range(0, len(b), 32)]


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
