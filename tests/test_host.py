"""CPU-side checks of the product package: constants against the golden vectors, the tuple <-> word
codec, the C ABI surface (every symbol of include/fourq_amd.h is exported and bound), and the
no-fallback rule (no GPU -> loud failure, never a CPU path).  No compute call is made without a GPU."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT, unhex
from fourq_amd import codec, constants


def _have_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def test_constants_match_reference(golden):
    kat = golden("kat.json", raw=True)
    assert constants.P127 == int(kat["p1271"], 16) and constants.N == int(kat["N"], 16)
    assert constants.d == unhex(kat["d"])
    assert (constants.Gx, constants.Gy) == unhex(kat["G"]) and (constants.Ox, constants.Oy) == unhex(kat["O"])
    import curve4q_oracle as o     # oracle is pinned to the reference by test_oracle_golden.py
    assert constants.ctau == o.ctau and constants.ctaudual == o.ctaudual and constants.cphi == o.cphi
    assert all(constants.cpsi[i] == o.cpsi[i] for i in (1, 2, 3, 4))
    assert constants.ELL == o.ELL and constants.BASIS == o.BASIS
    assert constants.OFFSET_C == o.OFFSET_C and constants.OFFSET_CP == o.OFFSET_CP


def test_generated_device_constants_are_current():
    """fourq_amd/csrc/constants.inc must be what tools/gen_constants.py produces from constants.py."""
    path = os.path.join(ROOT, "fourq_amd", "csrc", "constants.inc")
    before = open(path).read()
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_constants.py")], check=True, capture_output=True)
    assert open(path).read() == before


def test_codec_roundtrip(golden):
    rows = golden("mul.json")["var"]
    pts = [r[1] for r in rows]
    arr = codec.pack_points(pts, 5)
    assert arr.shape == (len(pts), 20) and arr.dtype == np.uint64
    assert codec.unpack_points(arr) == pts
    ms = [r[0] for r in rows] + [0, (1 << 256) - 1]
    assert codec.unpack_scalars(codec.pack_scalars(ms)) == ms
    with pytest.raises(ValueError):
        codec.pack_scalars([1 << 256])
    with pytest.raises(ValueError):
        codec.pack_scalars([-1])
    T = golden("tables.json")["tables"][0][1]
    assert codec.unpack_table(codec.pack_table(T)) == list(T)
    assert codec.unpack_fp2s(codec.pack_fp2s([(constants.P127, constants.P127 + 5)])) == ((0, 5),)   # reduced like `% p1271`
    with pytest.raises(ValueError):
        codec.pack_points([((0, 0), (1, 0))], 5)


def test_abi_exports_every_declared_symbol():
    """include/fourq_amd.h is the contract: each declared function must be exported by the built
    library and bound by the ctypes layer with the same arity."""
    from fourq_amd import _lib
    header_raw = open(os.path.join(ROOT, "include", "fourq_amd.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header_raw, flags=re.S)
    decls = re.findall(r"\b(?:int|const char \*|void)\s*\*?\s*(fourq_\w+)\s*\(([^;{]*)\)\s*;", header)
    names = {n for n, _ in decls}
    assert len(names) >= 30
    assert names == set(_lib.PROTOTYPES), names ^ set(_lib.PROTOTYPES)
    lib = _lib.load()
    for name, args in decls:
        assert hasattr(lib, name), name
        n_args = 0 if args.strip() == "void" else len([a for a in args.split(",") if a.strip()])
        assert len(_lib.PROTOTYPES[name][1]) == n_args, name
    assert lib.fourq_version() == _lib.ABI_VERSION == 300
    assert lib.fourq_strerror(-2).decode() == "no usable gfx950 HIP device"
    # enum values used from Python agree with the header
    for key, val in re.findall(r"FOURQ_(\w+)\s*=\s*(\d+)", header):
        if key in _lib.PRIM:
            assert _lib.PRIM[key] == int(val), key
    # ... and so do the sizes, limits and status codes the Python layer hard-codes
    defines = dict(re.findall(r"#define\s+(FOURQ_\w+)\s+\(?([-\w* ]+?)\)?\s*(?:/\*|$)", header_raw, flags=re.M))
    value = lambda k: eval(defines[k].replace("u", ""), {})
    assert value("FOURQ_COMB_POINTS") == _lib.COMB_POINTS and value("FOURQ_COMB_WORDS") == _lib.COMB_WORDS
    assert value("FOURQ_MAX_BATCH") == _lib.MAX_BATCH and value("FOURQ_BYTES_DECODE_BASE") == _lib.BYTES_DECODE_BASE
    for name in ("OK", "ERR_INVALID", "ERR_NODEVICE", "ERR_NOMEM", "ERR_HIP", "DH_OK", "DH_NOT_ON_CURVE", "DH_NEUTRAL",
                 "DECODE_OK", "DECODE_RESERVED_BIT", "DECODE_NOT_ON_CURVE", "DECODE_REF_ATTRIBUTE_ERROR"):
        assert value("FOURQ_" + name) == getattr(_lib, name), name
    import ctypes
    assert ctypes.sizeof(_lib.HostStats) == 48                     # struct fourq_host_stats: 2 doubles, 2 u64, u32 + 2 ints
    iw, ow = ctypes.c_size_t(), ctypes.c_size_t()
    for key, code in _lib.PRIM.items():
        assert lib.fourq_prim_words(code, ctypes.byref(iw), ctypes.byref(ow)) == 0 and iw.value and ow.value, key
    assert lib.fourq_prim_words(999, ctypes.byref(iw), ctypes.byref(ow)) == _lib.ERR_INVALID


def test_library_is_gfx950_only_and_links_no_oracle():
    """The shipped object holds a gfx950 code object and references nothing under oracle/."""
    from fourq_amd import _lib
    blob = open(_lib.LIB_PATH, "rb").read()
    assert b"gfx950" in blob
    assert b"fqo_" not in blob and b"fourq_oracle" not in blob
    for mod in ("engine.py", "curve4q.py", "fields.py", "codec.py", "_lib.py", "dist.py", "__init__.py", "constants.py", "build.py"):
        text = open(os.path.join(ROOT, "fourq_amd", mod)).read()
        assert "oracle" not in text.replace("no oracle", ""), mod


@pytest.mark.skipif(_have_gpu(), reason="checks the behaviour on a box WITHOUT a GPU")
def test_no_gpu_means_loud_failure_not_fallback():
    import fourq_amd
    from fourq_amd import curve4q
    with pytest.raises(fourq_amd.FourQError, match="no usable gfx950 HIP device"):
        fourq_amd.Engine(0)
    G1 = curve4q.AffineToR1(curve4q.Gx, curve4q.Gy)
    with pytest.raises(fourq_amd.FourQError):
        curve4q.MUL_endo(5, G1)
    with pytest.raises(ValueError):
        curve4q.MUL_endo(5, (curve4q.Gx, curve4q.Gy))      # shape check precedes any device work (curve4q.py:407)
    with pytest.raises(ValueError):
        curve4q.MUL_endo(1 << 256, G1)
    # the multi-device front end is as loud: no device, no engine -- and an explicit device list fails on its first context
    assert fourq_amd.device_count() == 0
    with pytest.raises(fourq_amd.FourQError, match="no usable gfx950 HIP device"):
        fourq_amd.MultiEngine()
    with pytest.raises(fourq_amd.FourQError, match="no usable gfx950 HIP device"):
        fourq_amd.MultiEngine([0, 0])


def test_null_arguments_are_rejected_without_a_device():
    import ctypes
    from fourq_amd import _lib
    lib = _lib.load()
    assert lib.fourq_ctx_create(0, None) == _lib.ERR_INVALID
    assert lib.fourq_ctx_destroy(None) == _lib.ERR_INVALID
    assert lib.fourq_mul_endo_batch(None, None, None, None, 0) == _lib.ERR_INVALID
    assert lib.fourq_dh_endo_batch(None, None, None, None, None, None, 0) == _lib.ERR_INVALID
    assert lib.fourq_prim_batch(None, 0, None, None, 0) == _lib.ERR_INVALID
