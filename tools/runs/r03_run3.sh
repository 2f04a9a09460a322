#!/bin/bash
# round 3, GPU call 3: where the queue kernel's time goes (pure fixed / pure variable batches), pair-lane microbenchmark
set -o pipefail
mkdir -p gpurun_out/r03
for n in 65536 131072; do for q in 0 1; do for ct in 0 1; do
  MIXED_N=$n FOURQ_MIXED_QUEUE=$q FOURQ_CT_SELECT=$ct python3 tools/mixed_probe.py 2>/dev/null | head -4 | tr '\n' ';'; echo
done; done; done | tee gpurun_out/r03/mixed_probe.txt
(cd tools/microbench && ./pairlane) | tee gpurun_out/r03/pairlane.txt
