#!/bin/bash
# the host-array pipeline (kernels + copies) at 2^20 elements against the number of streams in use in the process, and the link's duplex behaviour
set -e
OUT=gpurun_out/r04pipe
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
for x in 0 1 2 3 4 6; do tools/microbench/link_duplex $x; done > $OUT/duplex.txt 2>&1
GPU_MAX_HW_QUEUES=8 tools/microbench/link_duplex 4 >> $OUT/duplex.txt 2>&1
cat $OUT/duplex.txt
echo "--- no torch streams" > $OUT/probe.txt
python tools/pipeline_probe.py 20 --no-link >> $OUT/probe.txt 2>&1
echo "--- two torch streams used (kernels only)" >> $OUT/probe.txt
python tools/pipeline_probe.py 20 --streams >> $OUT/probe.txt 2>&1
echo "--- torch link test first" >> $OUT/probe.txt
python tools/pipeline_probe.py 20 >> $OUT/probe.txt 2>&1
echo "--- torch link test first, GPU_MAX_HW_QUEUES=8" >> $OUT/probe.txt
GPU_MAX_HW_QUEUES=8 python tools/pipeline_probe.py 20 >> $OUT/probe.txt 2>&1
grep -v amdgpu.ids $OUT/probe.txt
