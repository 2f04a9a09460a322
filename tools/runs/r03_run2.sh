#!/bin/bash
# round 3, GPU call 2: suite (mixed batches now run the persistent queue kernel), sign select behind the doubling A/B, queue kernel A/B
set -o pipefail
mkdir -p gpurun_out/r03
python -m pytest tests -m gpu -x -q > gpurun_out/r03/pytest2.log 2>&1 || { tail -30 gpurun_out/r03/pytest2.log; exit 1; }
tail -2 gpurun_out/r03/pytest2.log
echo "## A = product (sign by masked v_bitop3 select, selects fenced behind the doubling), B = round 2 (sign by address)" > gpurun_out/r03/ab_sign2.txt
tools/ab_bench.sh fourq_amd/libfourq_amd.so variants/libsignaddr.so cfg2 cfg3 cfg4 cfg5 >> gpurun_out/r03/ab_sign2.txt 2>&1 || { tail gpurun_out/r03/ab_sign2.txt; exit 1; }
cat gpurun_out/r03/ab_sign2.txt
B="python3 bench.py --workload cfg5 --no-configs --no-cpu-baseline --no-pcie --no-ct --no-alongside"
for rep in 1 2; do
  for ct in 0 1; do
    for q in 0 1; do
      FOURQ_CT_SELECT=$ct FOURQ_MIXED_QUEUE=$q $B 2>/dev/null | python3 -c "import json,sys; l=json.loads(sys.stdin.read()); print('cfg5 ct=$ct queue=$q  %.4f ms  parity %s' % (l['ms_per_step'], l['parity']['ok']))" || exit 1
    done
  done
done | tee gpurun_out/r03/mixed_queue.txt
