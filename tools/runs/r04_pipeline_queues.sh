#!/bin/bash
# The host-array pipeline against the number of hardware queues: a 2^20-element MUL_endo call on pinned arrays with nothing else in the
# process, with two more torch streams in use, and after torch copies on two streams; GPU_MAX_HW_QUEUES = 4 (the runtime's default),
# and the package's own default (8, set at import when the variable is unset).  -> profiles/r04_pipeline_queues.txt
set -e
OUT=gpurun_out/r04pipe
mkdir -p $OUT
R=$OUT/queues.txt
{
echo "== the link, raw HIP (tools/microbench/link_duplex.hip): 64 MiB each way on two streams"
tools/microbench/link_duplex 0
for q in 4 default; do
  if [ $q = default ]; then unset GPU_MAX_HW_QUEUES; echo "== GPU_MAX_HW_QUEUES unset: the package sets 8 at import"; else export GPU_MAX_HW_QUEUES=$q; echo "== GPU_MAX_HW_QUEUES=$q (what the runtime does by itself)"; fi
  echo "-- nothing else in the process"
  python tools/pipeline_probe.py 20 --no-link 2>&1 | grep -v amdgpu.ids
  echo "-- two more torch streams in use (a trivial kernel each)"
  python tools/pipeline_probe.py 20 --streams 2>&1 | grep -v amdgpu.ids
  echo "-- torch copies on two torch streams first (pinned memory both ways)"
  python tools/pipeline_probe.py 20 2>&1 | grep -v amdgpu.ids
done
} > $R 2>&1
cat $R
