#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r03
python -m pytest tests -m gpu -x -q > gpurun_out/r03/pytest5.log 2>&1 || { tail -40 gpurun_out/r03/pytest5.log; exit 1; }
tail -2 gpurun_out/r03/pytest5.log
B="python3 bench.py --workload cfg5 --no-configs --no-cpu-baseline --no-pcie --no-ct --no-alongside"
for rep in 1 2; do
  for ct in 0 1; do
      FOURQ_CT_SELECT=$ct $B 2>/dev/null | python3 -c "import json,sys; l=json.loads(sys.stdin.read()); print('cfg5 ct=$ct  %.4f ms  parity %s' % (l['ms_per_step'], l['parity']['ok']))" || exit 1
  done
done | tee gpurun_out/r03/mixed_ct_tail.txt
