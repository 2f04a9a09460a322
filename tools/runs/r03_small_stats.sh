#!/bin/bash
# rocprofv3 kernel statistics of the small-batch kernels (tools/ct_probe.py launches every hot entry point 4 times): four lanes per element at
# 4 096 elements, two at 20 000, both selection modes.  -> gpurun_out/r03s/<mode>_<n>/.../*_kernel_stats.csv
set -o pipefail
ROOTDIR=$(pwd); export TMPDIR=/tmp
mkdir -p $ROOTDIR/gpurun_out/r03s
cd /tmp
for mode in default ct; do
  for n in 4096 20000; do
    rocprofv3 --kernel-trace --stats --output-format csv -d $ROOTDIR/gpurun_out/r03s/${mode}_$n -- python3 $ROOTDIR/tools/ct_probe.py --mode $mode --scalars random --n $n --reps 20 > $ROOTDIR/gpurun_out/r03s/${mode}_$n.log 2>&1 || { tail -5 $ROOTDIR/gpurun_out/r03s/${mode}_$n.log; exit 1; }
  done
done
cd $ROOTDIR
for d in gpurun_out/r03s/*/; do echo "== $d"; cat $d/*/*_kernel_stats.csv | cut -d, -f1-8 | cut -c1-200 | sed -n 1,8p; done
