set -o pipefail
mkdir -p gpurun_out/r03c
CT_MODES=ct bash tools/ct_invariance.sh 4096 > gpurun_out/r03c/quad.txt 2>&1 || { tail -20 gpurun_out/r03c/quad.txt; exit 1; }
CT_MODES=ct bash tools/ct_invariance.sh 20000 > gpurun_out/r03c/pair.txt 2>&1 || { tail -20 gpurun_out/r03c/pair.txt; exit 1; }
grep -c "comb_quad_kernel" gpurun_out/r03c/quad.txt gpurun_out/r03c/pair.txt
