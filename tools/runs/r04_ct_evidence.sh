#!/bin/bash
# counter invariance of the constant-time kernels, final build: 2^16 elements both modes (+ the reversed class order, durations only), then the small-batch kernels
set -o pipefail
mkdir -p gpurun_out/r04c_ct
bash tools/ct_invariance.sh 65536 > gpurun_out/r04c_ct/full.txt 2>&1 || { tail -20 gpurun_out/r04c_ct/full.txt; exit 1; }
CT_CLASSES="same ones zero random" CT_TRACE_ONLY=1 bash tools/ct_invariance.sh 65536 > gpurun_out/r04c_ct/reversed.txt 2>&1 || { tail -20 gpurun_out/r04c_ct/reversed.txt; exit 1; }
CT_MODES=ct bash tools/ct_invariance.sh 4096 > gpurun_out/r04c_ct/quad.txt 2>&1 || { tail -20 gpurun_out/r04c_ct/quad.txt; exit 1; }
CT_MODES=ct bash tools/ct_invariance.sh 20000 > gpurun_out/r04c_ct/pair.txt 2>&1 || { tail -20 gpurun_out/r04c_ct/pair.txt; exit 1; }
grep -c "0.00 %" gpurun_out/r04c_ct/full.txt gpurun_out/r04c_ct/quad.txt gpurun_out/r04c_ct/pair.txt
