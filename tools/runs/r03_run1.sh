#!/bin/bash
# round 3, GPU call 1: suite on the new default (masked sign select), A/B against round 2's address choice, the cliff on both routes
set -o pipefail
mkdir -p gpurun_out/r03
python -m pytest tests -m gpu -x -q > gpurun_out/r03/pytest1.log 2>&1 || { tail -30 gpurun_out/r03/pytest1.log; exit 1; }
tail -2 gpurun_out/r03/pytest1.log
echo "## A = product (sign by masked select, v_bitop3), B = round 2 (sign by address)" > gpurun_out/r03/ab_sign.txt
tools/ab_bench.sh fourq_amd/libfourq_amd.so variants/libsignaddr.so >> gpurun_out/r03/ab_sign.txt 2>&1 || { tail gpurun_out/r03/ab_sign.txt; exit 1; }
cat gpurun_out/r03/ab_sign.txt
echo "## A = product (v_bitop3 select), B = xor-form select" > gpurun_out/r03/ab_selxor.txt
tools/ab_bench.sh fourq_amd/libfourq_amd.so variants/libselxor.so cfg2 cfg5 >> gpurun_out/r03/ab_selxor.txt 2>&1 || exit 1
cat gpurun_out/r03/ab_selxor.txt
NS=65536,65792,69632,73728,81920,98304,114688,131072,131328,163840,196608
echo "## default routes" > gpurun_out/r03/cliff.txt
python3 tools/perf_probe.py --modes endo_var,dh_endo,win_var --ns $NS 2>/dev/null | grep "n=" >> gpurun_out/r03/cliff.txt
echo "## FOURQ_SPLIT_MIN=65537 FOURQ_SPLIT_ENDO_MIN=65537 (two-kernel route from one element past a fused generation)" >> gpurun_out/r03/cliff.txt
FOURQ_SPLIT_MIN=65537 FOURQ_SPLIT_ENDO_MIN=65537 python3 tools/perf_probe.py --modes endo_var,dh_endo,win_var --ns $NS 2>/dev/null | grep "n=" >> gpurun_out/r03/cliff.txt
cat gpurun_out/r03/cliff.txt
