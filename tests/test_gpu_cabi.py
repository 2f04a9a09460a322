"""The C ABI from C: include/fourq_amd.h compiles as C99 (CPU test) and a plain C host program linked against
libfourq_amd.so reproduces the oracle's MUL_endo / DH_endo outputs bit for bit (GPU test).  This is the binding a
maintainer of another host language would write."""
import os
import random
import shutil
import subprocess

import numpy as np
import pytest

import curve4q_oracle as o
import oracle_c as oc
from conftest import ROOT
from fourq_amd import codec
from fourq_amd.build import build_library

HEADER_DIR = os.path.join(ROOT, "include")
SRC = os.path.join(ROOT, "tests", "c", "cabi_check.c")


@pytest.mark.skipif(shutil.which("gcc") is None, reason="needs a C compiler")
def test_header_is_plain_c99_and_cxx(tmp_path):
    probe = tmp_path / "probe.c"
    probe.write_text('#include "fourq_amd.h"\nint main(void) { return FOURQ_TABLE_WORDS == 128 && sizeof(fourq_host_stats) == 64 ? 0 : 1; }\n')
    subprocess.run(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-I", HEADER_DIR, "-c", "-o", str(tmp_path / "probe.o"), str(probe)], check=True)
    subprocess.run(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-I", HEADER_DIR, "-fsyntax-only", SRC], check=True)
    if shutil.which("g++"):
        subprocess.run(["g++", "-std=c++11", "-Wall", "-Werror", "-I", HEADER_DIR, "-x", "c++", "-fsyntax-only", str(probe)], check=True)


@pytest.mark.gpu
@pytest.mark.skipif(shutil.which("gcc") is None, reason="needs a C compiler")
def test_c_host_program_matches_the_oracle(tmp_path):
    lib = build_library()
    exe = str(tmp_path / "cabi_check")
    libdir = os.path.dirname(lib)
    subprocess.run(["gcc", "-std=c99", "-O1", "-I", HEADER_DIR, "-o", exe, SRC, "-L", libdir, "-lfourq_amd", "-Wl,-rpath," + libdir], check=True)
    n = 300
    rng = random.Random(4711)
    s = np.frombuffer(rng.getrandbits(256 * n).to_bytes(32 * n, "little"), dtype="<u8").reshape(n, 4).copy()
    G1 = codec.pack_point(o.AffineToR1(o.Gx, o.Gy))
    te = oc.table(oc.ENDO, G1)
    k = np.frombuffer(rng.getrandbits(256 * n).to_bytes(32 * n, "little"), dtype="<u8").reshape(n, 4).copy()
    pts = oc.mul(oc.ENDO, k, None, te)                                   # projective N-torsion points
    want = oc.mul(oc.ENDO, s, pts)
    g = np.repeat(codec.pack_point((o.Gx, o.Gy)).reshape(1, 8), n, axis=0)
    aff, st0 = oc.dh(oc.ENDO, k, g)
    aff[3] = 0                                                          # one point off the curve
    want_dh, want_st = oc.dh(oc.ENDO, s, aff)
    assert want_st[3] == 1 and not st0.any()
    path = tmp_path / "vectors.bin"
    with open(path, "wb") as fh:
        fh.write(np.uint64(n).tobytes())
        for a in (s, pts, want, aff, want_dh):
            fh.write(np.ascontiguousarray(a, dtype="<u8").tobytes())
        fh.write(want_st.astype(np.uint8).tobytes())
    env = dict(os.environ)
    # the C program brings no PyTorch: the library then binds the system HIP runtime (/opt/rocm/lib)
    env["LD_LIBRARY_PATH"] = "/opt/rocm/lib:" + env.get("LD_LIBRARY_PATH", "")
    proc = subprocess.run([exe, str(path)], capture_output=True, text=True, env=env)
    assert proc.returncode == 0, proc.stdout + proc.stderr
    assert "bit-exact through the C ABI" in proc.stdout


@pytest.mark.gpu
def test_engine_on_the_gpu_box_runs_the_build_of_these_sources():
    """The snapshot that travels to the GPU box carries the .so built in the container; a header edited without a rebuild would be
    tested and timed silently.  The id compiled into the library is a hash of the sources: compare them."""
    from fourq_amd import Engine, _lib, build
    e = Engine(0)
    try:
        assert e.build_id == build.source_id()
        assert _lib.build_matches_sources() is True
    finally:
        e.close()


@pytest.mark.gpu
def test_clock_probe_reads_a_plausible_shader_clock_idle_and_under_load():
    """fourq_diag_clock (round 5): 16 probe waves time a window of the 100 MHz counter in shader cycles.  Idle the chip may sit anywhere
    between its floor and its peak; with the context's stream kept busy by back-to-back MUL_endo launches it must hold a clock in the
    band MI355X holds under integer load, and the probe must not disturb the results being computed beside it."""
    import torch
    from bench import seeded_scalars
    from fourq_amd import Engine, codec, constants
    dev = torch.device("cuda", 0)
    g1 = codec.pack_point((constants.Gx, constants.Gy, (1, 0), constants.Gx, constants.Gy))
    with Engine(0) as e:
        idle = e.diag_clock(2000)
        assert 100 < idle["mhz_min"] <= idle["mhz"] <= idle["mhz_max"] < 2600, idle
        assert idle["under_load"] is False                      # an idle stream must not pass for a loaded one
        n = e.lanes
        te = e.table_endo(g1)
        s_h = seeded_scalars(8101, n)
        p_h = e.mul_endo_fixed(seeded_scalars(8102, n), te)
        s, p = (torch.from_numpy(a.view(np.int64)).to(dev) for a in (s_h, p_h))
        out = torch.empty((n, 20), dtype=torch.int64, device=dev)
        for _ in range(300):                                    # ~90 ms of work queued: the probe's 20 ms window lies inside it
            e.mul_endo_dev(s, p, out, n)
        busy = e.diag_clock(20000)
        assert busy["under_load"] is True
        e.sync()
        assert 1500 < busy["mhz_min"] <= busy["mhz"] <= busy["mhz_max"] < 2600, busy
        assert busy["mhz_max"] - busy["mhz_min"] < 200, busy     # the sixteen probes (two per XCD) agree
        assert np.array_equal(out.cpu().numpy().view(np.uint64), oc.mul(oc.ENDO, s_h, p_h))
        # round 6: the bracket form -- two stamp launches on the engine's stream around the work, paired per CU; its window is the work's span
        for _ in range(50):
            e.mul_endo_dev(s, p, out, n)
        e.sync()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        stream = torch.cuda.Stream(device=dev)
        e.set_stream(stream)
        e.diag_clock_begin()
        ev0.record(stream)
        for _ in range(100):
            e.mul_endo_dev(s, p, out, n)
        ev1.record(stream)
        e.diag_clock_stop()
        br = e.diag_clock_end()
        span_us = ev0.elapsed_time(ev1) * 1e3
        assert 1500 < br["mhz_min"] <= br["mhz"] <= br["mhz_max"] < 2600 and br["mhz_max"] - br["mhz_min"] < 250, br
        assert abs(br["window_us"] - span_us) < 0.02 * span_us + 100, (br, span_us)
        # the two forms agree on the clock under the same load -- to within the governor's ramp: the first probe's window was the first 20 ms
        # after idle (profiles/clock_ramp_r01.txt), the bracket's lies 100 ms into the load
        assert busy["mhz"] * 0.97 < br["mhz"] < busy["mhz"] * 1.2, (br, busy)
        assert np.array_equal(out.cpu().numpy().view(np.uint64), oc.mul(oc.ENDO, s_h, p_h))
        e.set_stream(None)


@pytest.mark.gpu
def test_clock_probe_says_so_when_the_load_did_not_outlast_its_window():
    """a reading of fourq_diag_clock is the clock UNDER LOAD only if the context's stream was still busy when the window closed.
    With a backlog shorter than the window (three launches, about a millisecond, against 20 ms) -- or a probe that queued behind the
    backlog on a shared hardware queue, which looks the same from here -- it reports under_load = 0 and bench.py's `after` mode marks
    the reading invalid.  (GPU_MAX_HW_QUEUES=1 does not reproduce the shared-queue case on this runtime: the probe's non-blocking stream
    still ran beside 200 queued launches and read 2 427 MHz; and hipStreamQuery is not the test -- it said "not ready" long after the stream had drained: an event recorded behind the backlog is.)"""
    import torch
    from bench import seeded_scalars
    from fourq_amd import Engine, codec, constants
    dev = torch.device("cuda", 0)
    g1 = codec.pack_point((constants.Gx, constants.Gy, (1, 0), constants.Gx, constants.Gy))
    with Engine(0) as e:
        n = e.lanes
        te = e.table_endo(g1)
        s = torch.from_numpy(seeded_scalars(8201, n).view(np.int64)).to(dev)
        p = torch.from_numpy(e.mul_endo_fixed(seeded_scalars(8202, n), te).view(np.int64)).to(dev)
        out = torch.empty((n, 20), dtype=torch.int64, device=dev)
        for _ in range(3):
            e.mul_endo_dev(s, p, out, n)
        short = e.diag_clock(20000)
        e.sync()
        assert short["under_load"] is False, short
        for _ in range(400):                                    # ~120 ms queued: the window lies inside it
            e.mul_endo_dev(s, p, out, n)
        long = e.diag_clock(20000)
        e.sync()
        assert long["under_load"] is True, long
