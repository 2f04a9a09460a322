#!/bin/bash
# config 5 through the persistent queue kernel (items over the two lists laid end to end, one mixed boundary item) against the
# compaction + prep + pointer-selected ladder route: same build, the route chosen by the test hook; both selection modes
set -o pipefail
OUT=gpurun_out/r04q
mkdir -p $OUT
export FOURQ_DEBUG_ROUTES=1
python -m pytest tests -m gpu -x -q -k "mixed" > $OUT/pytest_mixed.txt 2>&1 || { tail -30 $OUT/pytest_mixed.txt; exit 1; }
tail -1 $OUT/pytest_mixed.txt
{
for rep in 1 2; do
  for q in 0 1; do
    for ct in 0 1; do
      FOURQ_MIXED_QUEUE=$q FOURQ_CT_SELECT=$ct python3 bench.py --workload cfg5 --no-cpu-baseline --no-ct 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg5 ct=$ct queue=$q  %.4f ms  parity %s' % (d['ms_per_step'], d['parity']['ok']))"
    done
  done
done
for q in 0 1; do for n in 65536 131072 262144 1048576; do FOURQ_MIXED_QUEUE=$q MIXED_N=$n python3 tools/mixed_probe.py 2>/dev/null | tr '\n' ';'; echo; done; done
} > $OUT/queue_ab.txt 2>&1
cat $OUT/queue_ab.txt
