"""The chunk plan of the host-array calls (fourq_amd/csrc/pipeline_plan.h, plain C++): compiled with g++ and checked on the CPU -- the
plan covers the batch exactly, starts and ends with one generation, never asks for a chunk whose last input byte arrives after the
kernels before it have finished (the model the header states), and respects the slot size."""
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

    def run(n, unit, b_in, b_out, ns, gens=0):
        out = subprocess.run([exe, str(n), str(unit), str(b_in), str(b_out), str(ns), str(gens)], check=True, capture_output=True, text=True).stdout
        return [tuple(int(x) for x in p.split(":")) for p in out.split()]
    return run


def gens_of(pieces, unit):
    return [m // unit for _, m in pieces if m % unit == 0]


def test_plans_of_the_formats_at_two_to_the_twenty(plan):
    n = 1 << 20
    # raw R1 in and out: the copy in of a generation takes nearly as long as its kernels -- one generation per chunk throughout
    assert gens_of(plan(n, LANES, 192, 160, 4.63), LANES) == [1] * 16
    # affine in / out (96 + 64 B): chunks grow from both ends
    assert gens_of(plan(n, LANES, 96, 64, 4.73), LANES) == [1, 2, 4, 5, 3, 1]
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
    assert gens[0] == 1 and gens[-1] == 1
    cap = max(1, min(GENS_MAX, SLOT_MAX // (unit * (b_in + b_out) + 1)))
    assert max(gens) <= cap
    # the model the header states, played through: copies in back to back at h per generation, a chunk's kernels start when its last byte
    # is there and the chunk before it is done, cost k per generation plus a fixed price per chunk; copies out behind them at d per
    # generation.  The plan must never lose against one generation per chunk, and must not stall the kernels by more than the margin.
    h, k, d, gap = b_in * unit / LINK, ns * unit, b_out * unit / LINK, 20e3          # nanoseconds

    def play(sizes, kernel_time):
        t_in = t_k = t_out = stall = 0.0
        for g in sizes:
            t_in += h * g
            stall += max(0.0, t_in - t_k) if t_k else 0.0
            t_k = max(t_k, t_in) + kernel_time * g + gap
            t_out = max(t_out, t_k) + d * g
        return t_out, stall
    planned, stall = play(gens, k)
    ones, _ = play([1] * whole, k)
    assert planned <= ones * 1.0005, (gens, planned, ones)
    if k / SAFETY >= h:                           # kernel-bound formats: even kernels faster by the whole margin wait for no copy in
        assert play(gens, k)[1] <= play([1] * whole, k)[1] + 1e-6, gens


def test_forced_uniform_shape_is_what_the_gpu_tests_count_on(plan):
    for gens in (1, 2, 3, 4, 7):
        p = plan(7 * LANES + 555, LANES, 192, 160, 4.63, gens)
        inner = 5
        assert len(p) == 1 + (inner + gens - 1) // gens + 1 + 1
        assert p[0][1] == LANES and p[-2][1] == LANES and p[-1][1] == 555
    assert plan(LANES + 5, LANES, 192, 160, 4.63, 2) == [(0, LANES), (LANES, 5)]
