#!/bin/bash
# counter invariance of the constant-time kernels for small batches: four lanes per element (4 096 elements) and two (20 000)
set -o pipefail
mkdir -p gpurun_out/r03q
CT_MODES=ct bash tools/ct_invariance.sh 4096 > gpurun_out/r03q/ct_inv_quad.txt 2>&1 || { tail -20 gpurun_out/r03q/ct_inv_quad.txt; exit 1; }
CT_MODES=ct bash tools/ct_invariance.sh 20000 > gpurun_out/r03q/ct_inv_pair.txt 2>&1 || { tail -20 gpurun_out/r03q/ct_inv_pair.txt; exit 1; }
grep "pair_kernel" gpurun_out/r03q/ct_inv_quad.txt gpurun_out/r03q/ct_inv_pair.txt | grep "duration\|INSTS_VALU"
