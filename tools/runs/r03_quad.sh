#!/bin/bash
# round 3: four lanes per element -- suite, the one/two/four-lane table, counter invariance of the constant-time quad kernels (4 096 elements)
set -o pipefail
mkdir -p gpurun_out/r03q
python -m pytest tests -m gpu -x -q > gpurun_out/r03q/pytest.log 2>&1 || { tail -40 gpurun_out/r03q/pytest.log; exit 1; }
tail -1 gpurun_out/r03q/pytest.log
python3 tools/quad_probe.py > gpurun_out/r03q/quad_probe.txt 2>gpurun_out/r03q/quad_probe.err || { tail gpurun_out/r03q/quad_probe.err; exit 1; }
cat gpurun_out/r03q/quad_probe.txt
python3 tools/single_call_probe.py > gpurun_out/r03q/single_call.txt 2>&1 || exit 1
cat gpurun_out/r03q/single_call.txt
CT_MODES=ct bash tools/ct_invariance.sh 4096 > gpurun_out/r03q/ct_inv_quad.txt 2>&1 || { tail -20 gpurun_out/r03q/ct_inv_quad.txt; exit 1; }
grep -c "pair_kernel" gpurun_out/r03q/ct_inv_quad.txt
