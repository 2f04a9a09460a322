# full GPU suite on the round-4 defaults (asm ladder bodies in the fused, LDS and mixed-batch ladders; fused route for every variable-base batch;
# code placement), the fixed-base DH ladders before / after, the bench line
mkdir -p gpurun_out/r04c
python -m pytest tests -m gpu -x -q > gpurun_out/r04c/pytest_gpu.log 2>&1; tail -3 gpurun_out/r04c/pytest_gpu.log
for rep in 1 2; do
  FOURQ_AMD_LIB=$PWD/variants/libnoplace.so python3 tools/perf_probe.py --modes dh_fixed,endo_fixed,win_fixed --sizes 20 2>/dev/null | sed "s/^/before (hipcc ladders, 4 waves per SIMD, DH flavours spilling)  /"
  python3 tools/perf_probe.py --modes dh_fixed,endo_fixed,win_fixed --sizes 20 2>/dev/null | sed "s/^/after  (asm bodies, 2 waves per SIMD, no scratch)              /"
done > gpurun_out/r04c/fixed_base_before_after.txt 2>&1
cat gpurun_out/r04c/fixed_base_before_after.txt
python bench.py --steps 20 --warmup 5 > gpurun_out/r04c/bench.json 2> gpurun_out/r04c/bench.err
python -c "
import json; d=json.loads(open('gpurun_out/r04c/bench.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step']); print({k:(v.get('ms_per_step'), v.get('value')) for k,v in d.get('configs',{}).items()}); print(d.get('parity')); print(d.get('ct_select',{}).get('ratios', d.get('ct_select')))"
