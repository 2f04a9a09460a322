"""ctypes binding of libfourq_amd.so (include/fourq_amd.h).  No fallback: a missing library or a
missing GPU is an error, never a silent CPU path."""
import ctypes
import os
from ctypes import POINTER, c_char_p, c_int, c_size_t, c_uint8, c_uint64, c_void_p

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("FOURQ_AMD_LIB") or os.path.join(HERE, "libfourq_amd.so")   # override: experiments only

# The host-array calls keep THREE streams busy at once (copy in, kernels, copy out).  The HIP runtime maps all the streams of a process
# onto GPU_MAX_HW_QUEUES hardware queues, 4 unless the variable says otherwise, and streams that share a queue take turns: with two
# more streams in use elsewhere in the process (say, two torch streams) a 2^20-element MUL_endo call goes from 5.9 ms to 9.8, with four
# more to 12.7 -- copy, kernel, copy strictly in series (profiles/r04_pipeline_queues.txt).  With 8 queues it stays at 5.9.  The runtime
# reads the variable at its FIRST call, not when it is loaded, so a default set here -- at import, before this package has touched the
# GPU -- still counts unless the process has already used HIP; a value the user has set is left alone.  A C host sets it in its
# environment.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

OK, ERR_INVALID, ERR_NODEVICE, ERR_NOMEM, ERR_HIP = 0, -1, -2, -3, -4
DH_OK, DH_NOT_ON_CURVE, DH_NEUTRAL = 0, 1, 2
DECODE_OK, DECODE_RESERVED_BIT, DECODE_NOT_ON_CURVE, DECODE_REF_ATTRIBUTE_ERROR = 0, 1, 2, 3

MAX_BATCH = 0xFFFFFF00
ABI_VERSION = 600                    # fourq_version(): 0.6.0 (round 6: fourq_diag_clock_{begin,stop,end}, under_load; 0.5.0: fourq_ctx_set_host_timing, fourq_diag_clock)
COMB_POINTS = 1024 + 80              # FOURQ_COMB_POINTS: the fast comb and the one the constant-time mode scans
COMB_WORDS = COMB_POINTS * 12        # FOURQ_COMB_WORDS
BYTES_DECODE_BASE = 16


class HostStats(ctypes.Structure):
    """struct fourq_host_stats"""
    _fields_ = [("h2d_ms", ctypes.c_double), ("d2h_ms", ctypes.c_double), ("h2d_bytes", c_uint64), ("d2h_bytes", c_uint64),
                ("chunks", ctypes.c_uint32), ("pinned_in", c_int), ("pinned_out", c_int), ("kernels_ms", ctypes.c_double), ("kernels_span_ms", ctypes.c_double),
                ("planned_kernel_ns_per_elem", ctypes.c_double), ("planned_link_in_gbs", ctypes.c_double), ("planned_link_out_gbs", ctypes.c_double),
                ("measured_kernel_ns_per_elem", ctypes.c_double), ("planned_from_measurement", c_int)]


u64p = POINTER(c_uint64)
u8p = POINTER(c_uint8)

# name -> (restype, argtypes); every symbol include/fourq_amd.h declares
PROTOTYPES = {
    "fourq_version": (c_int, []),
    "fourq_build_id": (c_char_p, []),
    "fourq_strerror": (c_char_p, [c_int]),
    "fourq_last_error": (c_char_p, [c_void_p]),
    "fourq_device_count": (c_int, [POINTER(c_int)]),
    "fourq_ctx_create": (c_int, [c_int, POINTER(c_void_p)]),
    "fourq_ctx_destroy": (c_int, [c_void_p]),
    "fourq_ctx_set_stream": (c_int, [c_void_p, c_void_p]),
    "fourq_ctx_sync": (c_int, [c_void_p]),
    "fourq_ctx_set_ct_select": (c_int, [c_void_p, c_int]),
    "fourq_ctx_get_ct_select": (c_int, [c_void_p, POINTER(c_int)]),
    "fourq_ctx_lanes": (c_int, [c_void_p, POINTER(c_size_t)]),
    "fourq_ctx_reserve": (c_int, [c_void_p, c_size_t]),
    "fourq_host_alloc": (c_int, [c_void_p, c_size_t, POINTER(c_void_p)]),
    "fourq_host_free": (c_int, [c_void_p, c_void_p]),
    "fourq_ctx_host_stats": (c_int, [c_void_p, c_void_p]),
    "fourq_ctx_host_stats_sized": (c_int, [c_void_p, c_void_p, c_size_t]),
    "fourq_ctx_host_chunk_stamps": (c_int, [c_void_p, ctypes.c_uint32, POINTER(ctypes.c_double)]),
    "fourq_ctx_set_host_timing": (c_int, [c_void_p, c_int]),
    "fourq_diag_clock": (c_int, [c_void_p, ctypes.c_uint32, POINTER(ctypes.c_double), POINTER(ctypes.c_double), POINTER(ctypes.c_double), POINTER(c_int)]),
    "fourq_diag_clock_begin": (c_int, [c_void_p]),
    "fourq_diag_clock_stop": (c_int, [c_void_p]),
    "fourq_diag_clock_end": (c_int, [c_void_p, POINTER(ctypes.c_double), POINTER(ctypes.c_double), POINTER(ctypes.c_double), POINTER(ctypes.c_double)]),
    "fourq_dev_alloc": (c_int, [c_void_p, c_size_t, POINTER(c_void_p)]),
    "fourq_dev_free": (c_int, [c_void_p, c_void_p]),
    "fourq_dev_upload": (c_int, [c_void_p, c_void_p, c_void_p, c_size_t]),
    "fourq_dev_download": (c_int, [c_void_p, c_void_p, c_void_p, c_size_t]),
    "fourq_table_windowed": (c_int, [c_void_p, c_void_p, c_void_p]),
    "fourq_table_endo": (c_int, [c_void_p, c_void_p, c_void_p]),
    "fourq_mul_endo_batch": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_size_t]),
    "fourq_mul_windowed_batch": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_size_t]),
    "fourq_mul_endo_batch_dev": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_size_t]),
    "fourq_mul_windowed_batch_dev": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_size_t]),
    "fourq_mul_endo_affine_batch": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_size_t]),
    "fourq_mul_windowed_affine_batch": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_size_t]),
    "fourq_mul_endo_affine_batch_dev": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_size_t]),
    "fourq_mul_windowed_affine_batch_dev": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_size_t]),
    "fourq_mul_endo_bytes_batch": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t]),
    "fourq_mul_windowed_bytes_batch": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t]),
    "fourq_mul_endo_bytes_batch_dev": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t]),
    "fourq_mul_windowed_bytes_batch_dev": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t]),
    "fourq_mul_endo_fixed_batch": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_size_t]),
    "fourq_mul_windowed_fixed_batch": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_size_t]),
    "fourq_mul_endo_fixed_batch_dev": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_size_t]),
    "fourq_mul_windowed_fixed_batch_dev": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_size_t]),
    "fourq_mul_endo_mixed_batch": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t]),
    "fourq_mul_endo_mixed_batch_dev": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t]),
    "fourq_dh_endo_batch": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t]),
    "fourq_dh_windowed_batch": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t]),
    "fourq_dh_endo_batch_dev": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t]),
    "fourq_dh_windowed_batch_dev": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t]),
    "fourq_dh_endo_bytes_batch": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t]),
    "fourq_dh_windowed_bytes_batch": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t]),
    "fourq_dh_endo_bytes_batch_dev": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t]),
    "fourq_dh_windowed_bytes_batch_dev": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t]),
    "fourq_dh_exchange_batch": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t]),
    "fourq_dh_exchange_batch_dev": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t]),
    "fourq_dh_exchange_comb_batch": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t]),
    "fourq_dh_exchange_comb_batch_dev": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t]),
    "fourq_comb_table": (c_int, [c_void_p, c_void_p, c_void_p]),
    "fourq_comb_stage": (c_int, [c_void_p, c_void_p]),
    "fourq_comb_mul_batch": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t]),
    "fourq_comb_mul_batch_dev": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t]),
    "fourq_encode_batch": (c_int, [c_void_p, c_void_p, c_void_p, c_size_t]),
    "fourq_decode_batch": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_size_t]),
    "fourq_encode_batch_dev": (c_int, [c_void_p, c_void_p, c_void_p, c_size_t]),
    "fourq_decode_batch_dev": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_size_t]),
    "fourq_prim_words": (c_int, [c_int, POINTER(c_size_t), POINTER(c_size_t)]),
    "fourq_prim_batch": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_size_t]),
}

# enum fourq_prim
PRIM = {
    "FP_ADD": 0, "FP_SUB": 1, "FP_MUL": 2, "FP_SQR": 3, "FP_NEG": 4, "FP_INV": 5, "FP_INVSQRT": 6, "FP_SELECT": 7,
    "FP2_ADD": 16, "FP2_SUB": 17, "FP2_MUL": 18, "FP2_SQR": 19, "FP2_NEG": 20, "FP2_CONJ": 21, "FP2_INV": 22, "FP2_SELECT": 23,
    "PT_DBL": 32, "PT_ADD": 33, "PT_ADD_CORE": 34, "PT_R1TOR2": 35, "PT_R1TOR3": 36, "PT_R2TOR4": 37,
    "PT_TAU": 38, "PT_TAU_DUAL": 39, "PT_UPSILON": 40, "PT_CHI": 41, "PT_PHI": 42, "PT_PSI": 43,
    "PT_ON_CURVE": 44, "PT_COFACTOR392": 45, "PT_R1TOAFFINE": 46,
    "SC_DECOMPOSE": 64, "SC_RECODE": 65, "SC_WINDOWED": 66,
}


class FourQError(RuntimeError):
    pass


_lib = None


def load():
    """Load the shared library and bind every declared symbol (no GPU is touched here)."""
    global _lib
    if _lib is not None:
        return _lib
    # PyTorch-ROCm wheels bundle their own HIP/HSA runtime under the SONAME libfourq_amd.so also
    # needs (libamdhip64.so.7).  One process must use ONE runtime: importing torch first makes the
    # dynamic loader resolve our dependency to the copy torch already mapped, so device pointers,
    # streams and RCCL communicators created by torch are valid in our calls.
    if os.environ.get("FOURQ_NO_TORCH") != "1":
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
    if not os.path.exists(LIB_PATH):
        # a fresh checkout: build in-tree once (hipcc cross-compiles, ~1 minute); never a CPU substitute
        try:
            from .build import build_library
            build_library()
        except Exception as exc:
            raise FourQError("libfourq_amd.so is not built (%s) and building it failed: %s. "
                             "Run `python -m fourq_amd.build`; there is no CPU fallback." % (LIB_PATH, exc))
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)   # AttributeError here = header / library mismatch
        fn.restype = res
        fn.argtypes = args
    if lib.fourq_version() != ABI_VERSION:       # e.g. a library built before the comb table grew: its buffers would not match ours
        raise FourQError("%s is version %d, this package expects %d: rebuild it (`python -m fourq_amd.build --force`)"
                         % (LIB_PATH, lib.fourq_version(), ABI_VERSION))
    _warn_if_not_built_from_these_sources(lib)
    _lib = lib
    return lib


def build_matches_sources(lib=None):
    """True when the loaded library was compiled from the sources beside it (fourq_build_id() == build.source_id()), False when it
    was not, None when that cannot be told (an experiment library named by FOURQ_AMD_LIB, or a tree without its csrc/)."""
    if os.environ.get("FOURQ_AMD_LIB"):
        return None
    try:
        from .build import source_id
        want = source_id()
    except OSError:
        return None
    return (lib or load()).fourq_build_id().decode() == want


def _warn_if_not_built_from_these_sources(lib):
    # is_stale() goes by mtimes, which a checkout or a copy to the GPU box rewrites; the id compiled into the library is a hash of the
    # translation units, headers and flags, so it says for certain whether what will be tested and timed is what the sources say
    if build_matches_sources(lib) is False:
        import warnings
        warnings.warn("%s (build %s) was not compiled from the sources in %s: run `python -m fourq_amd.build` -- results and timings "
                      "below are those of the OLD code" % (LIB_PATH, lib.fourq_build_id().decode(), os.path.join(HERE, "csrc")), RuntimeWarning, stacklevel=3)


def check(rc, ctx=None):
    if rc == OK:
        return
    lib = load()
    msg = lib.fourq_strerror(rc).decode()
    if ctx is not None and rc == ERR_HIP:
        msg += ": " + lib.fourq_last_error(ctx).decode()
    raise FourQError("fourq_amd: %s (code %d)" % (msg, rc))
