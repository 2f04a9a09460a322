#!/bin/bash
# Runs on the GPU box (via gpurun): bench line, rocprofv3 kernel trace + stats, and separate PMC passes.
#   tools/profile_bench.sh <tag>      -> gpurun_out/<tag>/...
set -o pipefail
TAG=${1:-r01}
ROOTDIR=$(pwd)
OUT=$ROOTDIR/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python3 bench.py > $OUT/bench.json 2> $OUT/bench.err || { tail -20 $OUT/bench.err; exit 1; }
cat $OUT/bench.json
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOTDIR/bench.py --no-cpu-baseline > $OUT/trace.log 2>&1 || { tail -20 $OUT/trace.log; exit 1; }
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 $ROOTDIR/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $OUT/pmc_fetch.log 2>&1 || { tail -20 $OUT/pmc_fetch.log; exit 1; }
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 $ROOTDIR/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $OUT/pmc_write.log 2>&1 || { tail -20 $OUT/pmc_write.log; exit 1; }
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --kernel-trace --output-format csv -d $OUT/pmc_sq -- python3 $ROOTDIR/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $OUT/pmc_sq.log 2>&1 || { tail -20 $OUT/pmc_sq.log; exit 1; }
cd $ROOTDIR
find $OUT -name '*.csv' | head -30
