#!/bin/bash
# Counter evidence for the constant-time selection mode (GPU box):   tools/ct_invariance.sh [n]   -> gpurun_out/ct_inv/...
# For each selection mode and each class of secret scalars (tools/ct_probe.py) three rocprofv3 passes: kernel trace (durations),
# and two PMC passes (the TCC counters do not fit one pass with the SQ group).  --pmc passes carry --kernel-trace only.
# Summarise with:  python3 tools/ct_invariance_summary.py gpurun_out/ct_inv > profiles/r03_ct_invariance.txt
set -o pipefail
N=${1:-65536}
CLASSES=${CT_CLASSES:-"random zero ones same"}          # CT_CLASSES / CT_TRACE_ONLY=1: order and passes (to tell a data effect from an order effect)
ROOTDIR=$(pwd); export TMPDIR=/tmp
OUT=$ROOTDIR/gpurun_out/ct_inv; rm -rf $OUT; mkdir -p $OUT
cd /tmp
for mode in ${CT_MODES:-ct default}; do
  for cls in $CLASSES; do
    ARGS="$ROOTDIR/tools/ct_probe.py --mode $mode --scalars $cls --n $N"
    rocprofv3 --kernel-trace --output-format csv -d $OUT/${mode}_${cls}_trace -- python3 $ARGS > $OUT/${mode}_${cls}_trace.log 2>&1 || { tail -5 $OUT/${mode}_${cls}_trace.log; exit 1; }
    [ -n "$CT_TRACE_ONLY" ] && { echo "$mode $cls done"; continue; }
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/${mode}_${cls}_fetch -- python3 $ARGS > $OUT/${mode}_${cls}_fetch.log 2>&1 || { tail -5 $OUT/${mode}_${cls}_fetch.log; exit 1; }
    rocprofv3 --pmc WRITE_SIZE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --kernel-trace --output-format csv -d $OUT/${mode}_${cls}_sq -- python3 $ARGS > $OUT/${mode}_${cls}_sq.log 2>&1 || { tail -5 $OUT/${mode}_${cls}_sq.log; exit 1; }
    echo "$mode $cls done"
  done
done
cd $ROOTDIR
python3 tools/ct_invariance_summary.py $OUT
