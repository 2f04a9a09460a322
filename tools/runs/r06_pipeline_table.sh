#!/bin/bash
# DESIGN.md section 11's table from ONE box and ONE build (its round-5 cells came from three runs): every I/O format at 2^20 elements on pinned
# arrays, best and median of 9, with the device-resident floor of each, for the shipped defaults and for the shapes they replaced.
export FOURQ_DEBUG_ROUTES=1
P="python tools/pipeline_probe.py 20 --no-link --formats=r1,affine,bytes,fixed --reps=9"
echo "--- DEFAULT (round 6: measured planner inputs, DP plan, fused affine-in / XYZ-out) + device-resident floors"
$P --floor 2>&1 | grep -v amdgpu.ids
echo "--- round 5: compiled-in planner inputs, lift kernel + R1 rows (FOURQ_PIPE_MEASURE=0 FOURQ_FUSED_IO=0)"
FOURQ_PIPE_MEASURE=0 FOURQ_FUSED_IO=0 $P 2>&1 | grep -v amdgpu.ids
echo "--- rounds 2-4 shape: 3 slots, 1 generation per chunk, host hands slots on (+ FOURQ_FUSED_IO=0)"
FOURQ_PIPE_SLOTS=3 FOURQ_PIPE_GENS=1 FOURQ_PIPE_HOST_WAIT=1 FOURQ_FUSED_IO=0 $P 2>&1 | grep -v amdgpu.ids
echo "--- DEFAULT again"
$P --floor 2>&1 | grep -v amdgpu.ids
echo "--- DEFAULT under fourq_ctx_set_host_timing (six events per chunk)"
$P --timing 2>&1 | grep -v amdgpu.ids
