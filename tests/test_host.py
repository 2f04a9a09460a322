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
    """fourq_amd/csrc/constants.inc must be what tools/gen_constants.py produces from constants.py.  The check writes nothing: the
    file is a build dependency, and a rewrite -- even of identical text -- would make build.is_stale() true and cost the next
    build_library() four translation units."""
    from fourq_amd import build
    path = os.path.join(ROOT, "fourq_amd", "csrc", "constants.inc")
    stamp, stale = os.stat(path).st_mtime_ns, build.is_stale()
    tool = os.path.join(ROOT, "tools", "gen_constants.py")
    assert subprocess.run([sys.executable, tool, "--check"], capture_output=True).returncode == 0
    assert subprocess.run([sys.executable, tool, "--stdout"], check=True, capture_output=True, text=True).stdout == open(path).read()
    # the rewriting mode leaves an up-to-date file alone too
    assert "unchanged" in subprocess.run([sys.executable, tool], check=True, capture_output=True, text=True).stdout
    assert os.stat(path).st_mtime_ns == stamp and build.is_stale() == stale


def test_generated_ladder_bodies_are_current():
    """fourq_amd/csrc/ladder_asm_gfx950.inc must be what tools/asmgen/gen_ladder_step.py emits (the hand-scheduled ladder bodies)."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "asmgen", "gen_ladder_step.py")], check=True, capture_output=True, text=True).stdout
    assert out == open(os.path.join(ROOT, "fourq_amd", "csrc", "ladder_asm_gfx950.inc")).read()


def test_every_included_file_is_a_tracked_dependency_of_the_build():
    """pair.hip.h was included by kernels.hip.h but missing from build.py's HEADERS, so an edit to it neither rebuilt the
    library nor changed fourq_build_id.  Every #include "..." reachable from the translation units must be in SOURCES + HEADERS."""
    import re
    from fourq_amd import build
    known = {os.path.normpath(os.path.join(build.SRC_DIR, f)) for f in build.SOURCES + build.HEADERS}
    todo, seen = [os.path.join(build.SRC_DIR, f) for f in build.SOURCES], set()
    while todo:
        path = os.path.normpath(todo.pop())
        if path in seen:
            continue
        seen.add(path)
        assert path in known, "%s is included by the build but not tracked in fourq_amd/build.py" % os.path.relpath(path, ROOT)
        for inc in re.findall(r'^\s*#\s*include\s+"([^"]+)"', open(path).read(), re.M):
            todo.append(os.path.join(os.path.dirname(path), inc))
    assert os.path.exists(build.PLACE_TOOL) and build.is_stale() in (True, False)


def test_every_environment_variable_the_library_reads_is_documented():
    """every FOURQ_* string in the built library is either the one product option the header documents or a test
    hook behind the FOURQ_DEBUG_ROUTES gate that tools/README.md lists (and route_env() is the only reader of those)."""
    import re
    from fourq_amd import _lib
    blob = open(_lib.LIB_PATH, "rb").read()
    in_lib = {m.decode() for m in re.findall(rb"FOURQ_[A-Z][A-Z0-9_]+", blob)}
    header = open(os.path.join(ROOT, "include", "fourq_amd.h")).read()
    readme = open(os.path.join(ROOT, "tools", "README.md")).read()
    hooks = {"FOURQ_BLOCKS_PER_CU", "FOURQ_SPLIT_MIN", "FOURQ_SPLIT_ALL", "FOURQ_SPLIT_ENDO_MIN", "FOURQ_SPLIT_CHUNK", "FOURQ_PAIR_MAX",
             "FOURQ_QUAD_MAX", "FOURQ_MIXED_QUEUE", "FOURQ_NORM_K", "FOURQ_HOST_BOUNCE", "FOURQ_HOST_ZERO_COPY",
             "FOURQ_PIPE_SLOTS", "FOURQ_PIPE_GENS", "FOURQ_PIPE_HOST_WAIT", "FOURQ_PIPE_HOST_POLL", "FOURQ_PIPE_MEASURE", "FOURQ_FUSED_IO"}
    env_like = {v for v in in_lib if not v.startswith(("FOURQ_ERR", "FOURQ_OK", "FOURQ_DH_", "FOURQ_DECODE", "FOURQ_FP", "FOURQ_PT", "FOURQ_MAX", "FOURQ_TABLE", "FOURQ_COMB_", "FOURQ_R"))}
    assert env_like == hooks | {"FOURQ_CT_SELECT", "FOURQ_DEBUG_ROUTES"}, sorted(env_like ^ (hooks | {"FOURQ_CT_SELECT", "FOURQ_DEBUG_ROUTES"}))
    assert "FOURQ_CT_SELECT" in header and "FOURQ_DEBUG_ROUTES" in header
    for v in hooks | {"FOURQ_CT_SELECT", "FOURQ_DEBUG_ROUTES"}:
        assert v in readme, v
    src = open(os.path.join(ROOT, "fourq_amd", "csrc", "fourq_amd.hip")).read()
    for v in hooks:
        assert 'route_env("%s")' % v in src and 'getenv("%s")' % v not in src, v


def test_import_asks_for_enough_hardware_queues_and_respects_the_users_choice():
    """the host-array pipeline needs its three streams on three hardware queues (fourq_amd/_lib.py, profiles/r04_pipeline_queues.txt):
    importing the package sets GPU_MAX_HW_QUEUES=8 when the variable is unset and leaves a value of the user's alone"""
    code = "import os, fourq_amd; print(os.environ['GPU_MAX_HW_QUEUES'])"
    for given, want in ((None, "8"), ("2", "2"), ("16", "16")):
        env = {k: v for k, v in os.environ.items() if k != "GPU_MAX_HW_QUEUES"}
        if given is not None:
            env["GPU_MAX_HW_QUEUES"] = given
        out = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=env, capture_output=True, text=True, check=True).stdout.strip()
        assert out == want, (given, out)
    for doc in ("INTEGRATION.md", "README.md", os.path.join("tools", "README.md")):
        assert "GPU_MAX_HW_QUEUES" in open(os.path.join(ROOT, doc)).read(), doc


def test_tuple_codec_is_fast_enough_to_be_worth_swapping_in():
    """the tuple-level API spent 4.5 + 5.6 us per element packing and unpacking in Python loops.  With
    csrc/fastcodec.c (built in-tree by build()) an R1 point + scalar packs and an R1 point unpacks in well under a microsecond on this
    container's CPU; the floor here is loose enough for a loaded box (best of 5) and tight enough to catch a fall back to Python."""
    import random
    import time
    assert codec._fc is not None, "fourq_amd/_fastcodec*.so is not built: run __graft_entry__.build()"
    rnd = random.Random(5)
    n = 20000
    pts = [tuple((rnd.randrange(constants.P127), rnd.randrange(constants.P127)) for _ in range(5)) for _ in range(n)]
    ms = [rnd.getrandbits(256) for _ in range(n)]
    pack = unpack = 1e9
    for _ in range(5):
        t = time.perf_counter(); arr = codec.pack_points(pts, 5); sc = codec.pack_scalars(ms); pack = min(pack, time.perf_counter() - t)
        t = time.perf_counter(); back = codec.unpack_points(arr); unpack = min(unpack, time.perf_counter() - t)
    assert back == pts and codec.unpack_scalars(sc) == ms
    assert pack / n < 1.5e-6 and unpack / n < 1.5e-6, (pack / n, unpack / n)


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
    assert lib.fourq_version() == _lib.ABI_VERSION == 600
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
    assert ctypes.sizeof(_lib.HostStats) == 104                    # struct fourq_host_stats: 2 doubles, 2 u64, u32 + 2 ints (+ pad), 2 doubles; 0.6.0: 4 doubles, int (+ pad)
    assert value("FOURQ_ABI_VERSION") == _lib.ABI_VERSION
    iw, ow = ctypes.c_size_t(), ctypes.c_size_t()
    for key, code in _lib.PRIM.items():
        assert lib.fourq_prim_words(code, ctypes.byref(iw), ctypes.byref(ow)) == 0 and iw.value and ow.value, key
    assert lib.fourq_prim_words(999, ctypes.byref(iw), ctypes.byref(ow)) == _lib.ERR_INVALID


def test_loaded_library_is_built_from_these_sources(tmp_path):
    """nothing tied the loaded .so to the sources at load time.  fourq_build_id() (a hash of the translation units,
    headers and flags, compiled in) must equal build.source_id() of the tree -- no GPU needed to ask -- and the loader warns when it
    does not."""
    import warnings
    from fourq_amd import _lib, build
    lib = _lib.load()
    assert lib.fourq_build_id().decode() == build.source_id(), "libfourq_amd.so is older than fourq_amd/csrc: python -m fourq_amd.build"
    assert _lib.build_matches_sources(lib) is True

    class Old:                                     # a library that answers with another id makes the loader speak up
        @staticmethod
        def fourq_build_id():
            return b"0123456789abcdef"
    with warnings.catch_warnings(record=True) as seen:
        warnings.simplefilter("always")
        _lib._warn_if_not_built_from_these_sources(Old)
    assert len(seen) == 1 and "was not compiled from the sources" in str(seen[0].message)
    with warnings.catch_warnings(record=True) as seen:
        warnings.simplefilter("always")
        _lib._warn_if_not_built_from_these_sources(lib)
    assert not seen


def test_package_version_is_the_abi_version():
    import fourq_amd
    from fourq_amd import _lib
    major, minor, patch = (int(x) for x in fourq_amd.__version__.split("."))
    assert 10000 * major + 100 * minor + patch == _lib.ABI_VERSION == _lib.load().fourq_version()


def test_a_codec_extension_that_does_not_build_does_not_stop_the_library_build(monkeypatch, capsys):
    """_fastcodec is optional (codec.py has a pure-Python path); build_library() must warn and go on."""
    from fourq_amd import build

    def broken(force=False, verbose=False):
        raise RuntimeError("no C compiler for fourq_amd/csrc/fastcodec.c")
    monkeypatch.setattr(build, "build_fastcodec", broken)
    monkeypatch.setattr(build, "is_stale", lambda: False)
    assert build.build_library() == build.LIB_PATH
    assert "falls back to its pure-Python conversions" in capsys.readouterr().err


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


def test_documents_quote_the_committed_bench_line():
    """DESIGN.md section 6 and README.md carry numbers generated from the driver's own record (BENCH_rNN.json.parsed, newest parsed one, stated
    first) and from profiles/r06_bench_driver_args_run*.full.json (the builder's runs of the driver's protocol); the generator's --check mode must
    find nothing to change, and neither document has a line a reviewer has to scroll sideways for.  The committed compact lines are what the
    driver reads: at most 6 144 bytes each, and bench.compact_line reproduces them from the committed full records."""
    tool = os.path.join(ROOT, "tools", "sync_design_numbers.py")
    proc = subprocess.run([sys.executable, tool, "--check"], capture_output=True, text=True)
    assert proc.returncode == 0, proc.stderr
    # ... and stays green when the driver drops a NEWER record into the tree after the last commit (it does, every round): the documents
    # name the record they quote, and the check compares them with THAT one
    import json as _json
    newest = sorted(f for f in os.listdir(ROOT) if re.fullmatch(r"BENCH_r\d+\.json", f))
    parsed = [f for f in newest if isinstance(_json.load(open(os.path.join(ROOT, f))).get("parsed"), dict)]
    if parsed:
        later = os.path.join(ROOT, "BENCH_r98.json")
        assert not os.path.exists(later)
        try:
            rec = _json.load(open(os.path.join(ROOT, parsed[-1])))
            rec["parsed"]["value"] = 1.0
            with open(later, "w") as fh:
                _json.dump(rec, fh)
            proc = subprocess.run([sys.executable, tool, "--check"], capture_output=True, text=True)
            assert proc.returncode == 0, proc.stderr
        finally:
            if os.path.exists(later):
                os.remove(later)
    for doc in ("DESIGN.md", "HISTORY.md"):
        long = [i + 1 for i, ln in enumerate(open(os.path.join(ROOT, doc), encoding="utf-8").read().split("\n")) if len(ln) > 200]
        assert not long, "%s: lines %s are longer than 200 characters" % (doc, long[:5])
    assert os.path.getsize(os.path.join(ROOT, "DESIGN.md")) < 48 * 1024       # the current design only; the rest is HISTORY.md
    import json
    import bench
    for k in (1, 2, 3):
        line = open(os.path.join(ROOT, "profiles", "r06_bench_driver_args_run%d.json" % k)).read().strip()
        with open(os.path.join(ROOT, "profiles", "r06_bench_driver_args_run%d.full.json" % k)) as fh:
            full = json.load(fh)
        assert len(line) <= bench.LINE_MAX_BYTES and "\n" not in line
        assert json.loads(bench.compact_line(full)) == json.loads(line)
        for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
                    "config", "roofline", "cpu_baseline"):
            assert key in json.loads(line), key


def test_compact_line_refuses_to_outgrow_the_driver():
    """the line grew round by round until the driver could not read it.  bench.compact_line takes numbers and short labels from
    the full record and REFUSES (SystemExit, no line at all) to print more than LINE_MAX_BYTES; strings are cut to their label length."""
    import json
    import bench
    with open(os.path.join(ROOT, "profiles", "r06_bench_driver_args_run1.full.json")) as fh:
        full = json.load(fh)
    line = bench.compact_line(full)
    assert len(line) < 3000 and json.loads(line)["roofline"]["frac"] == full["roofline"]["frac"]
    bloated = json.loads(json.dumps(full))
    bloated["cpu_baseline"]["sample"] = "x" * 5000                       # prose creeping back in: cut to 160 characters
    assert len(json.loads(bench.compact_line(bloated))["cpu_baseline"]["sample"]) == 160
    bloated["configs"] = {"cfg%d" % k: bloated["configs"]["cfg3"] for k in range(3, 60)}      # ... or records: refused
    with pytest.raises(SystemExit, match="over the"):
        bench.compact_line(bloated)
