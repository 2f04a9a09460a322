# same-box: the fused route (asm ladder bodies) against the two-kernel route for DH batches (cfg4) and large MUL_windowed / DH batches
mkdir -p gpurun_out/r04b
export FOURQ_DEBUG_ROUTES=1      # FOURQ_SPLIT_MIN below is a test hook
for rep in 1 2; do
  for mode in default nosplit; do
    if [ $mode = nosplit ]; then export FOURQ_SPLIT_MIN=1000000000; else unset FOURQ_SPLIT_MIN; fi
    python3 bench.py --workload cfg4 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg4 $mode', d['ms_per_step'])"
    python3 tools/perf_probe.py --modes win_var,dh_endo --sizes 17,18,20 2>/dev/null | sed "s/^/$mode /"
  done
done
