#!/usr/bin/env python3
"""Where every chunk of ONE host-array call sits in time (fourq_ctx_set_host_timing + fourq_ctx_host_chunk_stamps): copy in, kernels, copy out per
chunk, the kernel stream's idle time in front of each chunk and what it was waiting for.  Used to find the cause of round 5's "fourth slot is
WORSE under the host's hand-over" (ADVICE r5): the same 2^20-element raw-R1 call with the GPU handing slots on (default),
and with the host handing them on with 3 and with 4 slots.

    python tools/pipeline_chunks.py [--format r1|affine|bytes] [--lg 20] [--detail]
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import numpy as np

from bench import seeded_scalars
from fourq_amd import Engine, codec, constants

FMT = next((sys.argv[i + 1] for i, a in enumerate(sys.argv) if a == "--format"), "r1")
LG = int(next((sys.argv[i + 1] for i, a in enumerate(sys.argv) if a == "--lg"), "20"))
DETAIL = "--detail" in sys.argv
n = 1 << LG
g1 = codec.pack_point((constants.Gx, constants.Gy, (1, 0), constants.Gx, constants.Gy))


def engine(**knobs):
    saved = {k: os.environ.get(k) for k in list(knobs) + ["FOURQ_DEBUG_ROUTES"]}
    os.environ.update({k: str(v) for k, v in knobs.items()}, FOURQ_DEBUG_ROUTES="1")
    try:
        return Engine(0)
    finally:
        for k, v in saved.items():
            os.environ.pop(k, None) if v is None else os.environ.__setitem__(k, v)


def run(label, **knobs):
    with engine(**knobs) as e:
        te = e.table_endo(g1)
        s = e.host_array(seeded_scalars(1, n))
        r1 = e.mul_endo_fixed(seeded_scalars(2, n), te)
        if FMT == "r1":
            p, o = e.host_array(r1), e.host_empty((n, 20))
            call = lambda: e.mul_endo(s, p, out=o)
        else:
            import oracle_c as oc
            aff = oc.r1_to_affine(r1)
            if FMT == "affine":
                p, o = e.host_array(aff), e.host_empty((n, 8))
                call = lambda: e.mul_affine(s, p, out=o)
            else:
                p, o, st = e.host_array(oc.encode(aff)), e.host_empty((n, 32), np.uint8), e.host_empty((n,), np.uint8)
                call = lambda: e.mul_bytes(s, p, out=o, status=st)
        for _ in range(12):
            call()
        ts = []
        for _ in range(9):
            t0 = time.perf_counter(); call(); ts.append((time.perf_counter() - t0) * 1e3)
        ts.sort()
        e.host_timing(True)
        call(); call()
        t0 = time.perf_counter(); call(); timed_ms = (time.perf_counter() - t0) * 1e3
        rows = e.host_chunk_stamps()
        st = e.host_stats()
        e.host_timing(False)
        idle = [rows[k][4] - rows[k - 1][5] for k in range(1, len(rows))]
        late = [rows[k][1] - rows[k - 1][5] for k in range(1, len(rows))]       # > 0: the chunk's last input byte arrived AFTER the previous chunk's kernels were done
        kern = [r[5] - r[4] for r in rows]
        print("%-34s untimed: best %.3f median %.3f ms | timed call %.3f ms, %d chunks, planned with %.2f ns/elem, link %.1f / %.1f GB/s (%s)" % (
            label, ts[0], ts[len(ts) // 2], timed_ms, len(rows), st["planned_kernel_ns_per_elem"], st["planned_link_in_gbs"], st["planned_link_out_gbs"],
            "measured" if st["planned_from_measurement"] else "guess"))
        print("    kernels: busy %.3f ms, span %.3f ms, idle between chunks %.3f ms (max %.0f us) | chunks whose input came late: %d (summed %.3f ms) | "
              "first kernel starts at %.3f ms, last copy out ends %.3f ms after the last kernel" % (
                  sum(kern), rows[-1][5] - rows[0][4], sum(idle), 1e3 * max(idle) if idle else 0, sum(1 for x in late if x > 0.005), sum(x for x in late if x > 0),
                  rows[0][4], rows[-1][3] - rows[-1][5]))
        if DETAIL:
            print("      k   copy-in            kernels            copy-out           idle-before  input-late")
            for k, (i0, i1, o0, o1, k0, k1) in enumerate(rows):
                print("    %3d   %6.3f .. %6.3f   %6.3f .. %6.3f   %6.3f .. %6.3f   %8.0f us  %8.0f us" % (
                    k, i0, i1, k0, k1, o0, o1, 1e3 * idle[k - 1] if k else 0, 1e3 * late[k - 1] if k else 0))


print("# %s I/O, 2^%d elements, pinned arrays; stamps of ONE call under fourq_ctx_set_host_timing (six events per chunk: the timed call is ~0.2 ms slower)" % (FMT, LG))
run("default: GPU hand-over, 4 slots")
run("host hand-over, 3 slots", FOURQ_PIPE_HOST_WAIT=1, FOURQ_PIPE_SLOTS=3)
run("host hand-over, 4 slots", FOURQ_PIPE_HOST_WAIT=1, FOURQ_PIPE_SLOTS=4)
run("host hand-over, 4 slots, polling", FOURQ_PIPE_HOST_WAIT=1, FOURQ_PIPE_SLOTS=4, FOURQ_PIPE_HOST_POLL=1)
run("host hand-over, 5 slots", FOURQ_PIPE_HOST_WAIT=1, FOURQ_PIPE_SLOTS=5)
run("host hand-over, 4 slots, 1 gen", FOURQ_PIPE_HOST_WAIT=1, FOURQ_PIPE_SLOTS=4, FOURQ_PIPE_GENS=1)
run("GPU hand-over, 3 slots", FOURQ_PIPE_SLOTS=3)
run("default, round-5 planner inputs", FOURQ_PIPE_MEASURE=0)
if FMT != "r1":
    run("round-5 route: lift + R1 rows", FOURQ_FUSED_IO=0)
