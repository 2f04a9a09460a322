#!/bin/bash
# does what else the process has used (torch streams, a second context) change the host-array pipeline?  bench.py's process has both.
set -e
OUT=gpurun_out/${1:-r05streams}
mkdir -p $OUT
: > $OUT/streams.txt
for flags in "" "--torch-stream" "--second-ctx" "--torch-stream --second-ctx" "--torch-stream --second-ctx-host" "--streams --torch-stream --second-ctx"; do
  echo "--- flags: $flags" >> $OUT/streams.txt
  python tools/pipeline_probe.py 20 --no-link --formats=r1,affine --reps=7 $flags 2>&1 | grep -v amdgpu.ids >> $OUT/streams.txt
done
echo "--- GPU_MAX_HW_QUEUES=16, flags: --torch-stream --second-ctx-host" >> $OUT/streams.txt
GPU_MAX_HW_QUEUES=16 python tools/pipeline_probe.py 20 --no-link --formats=r1,affine --reps=7 --torch-stream --second-ctx-host 2>&1 | grep -v amdgpu.ids >> $OUT/streams.txt
cat $OUT/streams.txt
