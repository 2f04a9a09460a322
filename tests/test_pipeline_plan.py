"""The chunk plan of the host-array calls (fourq_amd/csrc/pipeline_plan.h, plain C++): compiled with g++ and checked on the CPU -- the
plan covers the batch exactly, respects the slot size, follows the rates it is given (the context's own measurements, round 6) and, under
the model the header states, never loses against one generation per chunk and stalls the kernels only where that saves chunk boundaries."""
import os
import subprocess

import pytest

from conftest import ROOT

LANES = 65536                      # one generation of the fused variable-base kernels on MI355X (256 CUs x 256 lanes)
LINK, SAFETY, SLOT_MAX, GENS_MAX = 48.0, 0.85, 64 << 20, 8


@pytest.fixture(scope="module")
def plan(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("plan") / "plan_dump")
    subprocess.run(["g++", "-std=c++17", "-O1", "-Wall", "-Werror", "-I", os.path.join(ROOT, "fourq_amd", "csrc"),
                    os.path.join(ROOT, "tests", "c", "plan_dump.cpp"), "-o", exe], check=True)

    def run(n, unit, b_in, b_out, ns, gens=0, link=None):
        args = [exe, str(n), str(unit), str(b_in), str(b_out), str(ns), str(gens)] + ([str(link[0]), str(link[1])] if link else [])
        out = subprocess.run(args, check=True, capture_output=True, text=True).stdout
        return [tuple(int(x) for x in p.split(":")) for p in out.split()]
    return run


def gens_of(pieces, unit):
    return [m // unit for _, m in pieces if m % unit == 0]


MARGIN, GAP = 0.95, 20e3         # pipeline_plan.h: MODEL_MARGIN, GAP_NS


def play(sizes, unit, b_in, b_out, ns, link=(LINK, LINK), margin=MARGIN):
    """pipeline_plan.h's model, restated: (time of the last output byte, summed stall of the kernel stream), nanoseconds"""
    h, k, d = b_in * unit / link[0] / margin, ns * unit, b_out * unit / link[1] / margin
    t_in = t_k = t_out = stall = 0.0
    for g in sizes:
        t_in += h * g
        stall += max(0.0, t_in - t_k) if t_k else 0.0
        t_k = max(t_k, t_in) + k * g + GAP
        t_out = max(t_out, t_k) + d * g
    return t_out, stall


def test_plans_of_the_formats_at_two_to_the_twenty(plan):
    n = 1 << 20
    # raw R1 in and out at round 5's guess of the kernel time: the copy in of a generation takes nearly as long as its kernels, so round 5 kept
    # one generation per chunk throughout; the dynamic program accepts a short stall where it saves a chunk boundary
    g = gens_of(plan(n, LANES, 192, 160, 4.63), LANES)
    assert g == [1, 1, 1, 1, 1, 1, 2, 2, 2, 2, 1, 1]
    assert play(g, LANES, 192, 160, 4.63)[0] < play([1] * 16, LANES, 192, 160, 4.63)[0] * 0.995      # four boundaries fewer for a 0.05 ms stall
    # ... and keeps one generation per chunk when the kernels are as fast as this round's box measured them (4.48 ns: k / h = 1.06)
    assert gens_of(plan(n, LANES, 192, 160, 4.48), LANES) == [1] * 16
    # constant-time kernels (x 1.25) or a faster link leave room for larger chunks: the plan follows its inputs
    g_ct = gens_of(plan(n, LANES, 192, 160, 4.48 * 1.25), LANES)
    assert len(g_ct) < 12 and g_ct[0] == 1 and g_ct[-1] == 1 and sum(g_ct) == 16
    assert len(gens_of(plan(n, LANES, 192, 160, 4.48, 0, (57.0, 57.0)), LANES)) < 16
    assert gens_of(plan(n, LANES, 192, 160, 4.48, 0, (40.0, 40.0)), LANES) == [1] * 16       # a slower link: the link is the pace
    # affine in / out (96 + 64 B): chunks grow from both ends
    g = gens_of(plan(n, LANES, 96, 64, 4.73), LANES)
    assert g[0] == 1 and g[-1] == 1 and sum(g) == 16 and len(g) <= 6
    # encoded points (64 + 33 B)
    g = gens_of(plan(n, LANES, 64, 33, 5.03), LANES)
    assert g[0] == 1 and g[-1] == 1 and sum(g) == 16 and len(g) <= 6
    # cfg3's call: scalars in, R1 out, generation = 131 072 elements of the two-waves-per-SIMD ladders
    g = gens_of(plan(n, 2 * LANES, 32, 160, 7.65), 2 * LANES)
    assert g[0] == 1 and g[-1] == 1 and sum(g) == 8 and max(g) <= 2           # 64 MiB slots: 2 x 131 072 x 192 B


@pytest.mark.parametrize("n", [LANES + 1, 2 * LANES, 2 * LANES + 1234, 7 * LANES + 555, 16 * LANES, 33 * LANES + 1, 1000 * LANES + 77])
@pytest.mark.parametrize("b_in,b_out,ns", [(192, 160, 4.63), (96, 64, 4.73), (64, 33, 5.03), (32, 160, 7.65), (32, 65, 1.09), (64, 65, 8.5), (64, 32, 0.05), (193, 160, 4.1)])
def test_plan_properties(plan, n, b_in, b_out, ns):
    unit = LANES
    pieces = plan(n, unit, b_in, b_out, ns)
    # exact cover, in order
    off = 0
    for o, m in pieces:
        assert o == off and m > 0
        off += m
    assert off == n
    whole = n // unit
    gens = [m // unit for _, m in pieces[:len(pieces) - (1 if n % unit else 0)]]
    assert all(m % unit == 0 for _, m in pieces[:len(gens)]) and sum(gens) == whole
    if n % unit:
        assert pieces[-1][1] == n % unit
    cap = max(1, min(GENS_MAX, SLOT_MAX // (unit * (b_in + b_out) + 1)))
    assert max(gens) <= cap and gens[0] <= 2 and gens[-1] <= 2
    # Under the header's model (copies in back to back, a chunk's kernels behind its last byte and the chunk before it, a fixed price per
    # chunk, copies out behind them) the plan never loses against one generation per chunk, nor against round 5's greedy ramps, and
    # stalls the kernel stream only where the stall is cheaper than the chunk boundaries it saves.
    planned, stall = play(gens, unit, b_in, b_out, ns)
    ones, stall_ones = play([1] * whole, unit, b_in, b_out, ns)
    assert planned <= ones * 1.0000001, (gens, planned, ones)
    assert stall <= stall_ones + (whole - len(gens)) * GAP + 1e-6, (gens, stall, stall_ones)


def test_forced_uniform_shape_is_what_the_gpu_tests_count_on(plan):
    for gens in (1, 2, 3, 4, 7):
        p = plan(7 * LANES + 555, LANES, 192, 160, 4.63, gens)
        inner = 5
        assert len(p) == 1 + (inner + gens - 1) // gens + 1 + 1
        assert p[0][1] == LANES and p[-2][1] == LANES and p[-1][1] == 555
    assert plan(LANES + 5, LANES, 192, 160, 4.63, 2) == [(0, LANES), (LANES, 5)]
