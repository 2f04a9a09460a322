for rep in 1 2; do
export FOURQ_DEBUG_ROUTES=1
for chunk in 131072 262144; do
  FOURQ_SPLIT_CHUNK=$chunk FOURQ_AMD_LIB=$PWD/variants/libpreasm.so python3 bench.py --workload cfg4 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('preasm chunk $chunk', d['ms_per_step'])"
done
FOURQ_AMD_LIB=$PWD/fourq_amd/libfourq_amd.so python3 bench.py --workload cfg4 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('product', d['ms_per_step'])"
done
