#!/bin/bash
# pageable callers of the host-array pipeline at 2^20 elements: the bounce path (round 5's shapes against rounds 2-4's), and the arrays handed
# to hipMemcpyAsync as they are
set -e
OUT=gpurun_out/${1:-r05pageable}
mkdir -p $OUT
export FOURQ_DEBUG_ROUTES=1
: > $OUT/pageable.txt
for env in "X=1" "FOURQ_PIPE_SLOTS=4" "FOURQ_PIPE_SLOTS=2" "FOURQ_HOST_BOUNCE=0" "X=2"; do
  echo "--- $env (caller's out= array reused)" >> $OUT/pageable.txt
  env $env python tools/pipeline_probe.py 20 --no-link --pageable --formats=r1,affine,fixed --reps=7 2>&1 | grep -v amdgpu.ids >> $OUT/pageable.txt
done
cat $OUT/pageable.txt
