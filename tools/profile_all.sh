#!/bin/bash
# Runs on the GPU box (via gpurun): for every BASELINE workload one bench line, one rocprofv3 kernel trace + stats and
# three separate PMC passes (FETCH_SIZE / WRITE_SIZE / SQ group: the TCC counters do not fit one pass).  A --pmc pass carries
# --kernel-trace only (needed for the per-kernel rows); it is never combined with --sys-trace / --runtime-trace or the
# hip / hsa / memory-copy / scratch-memory / marker domains, which this pool's gpurun refuses (validated on rocprofv3 of ROCm 7.2).
#   tools/profile_all.sh <tag> [workloads...]      -> gpurun_out/<tag>/<workload>/...
#   python tools/summarize_profiles.py gpurun_out/<tag> <tag>    (afterwards, anywhere)
set -o pipefail
TAG=${1:-r03}; shift
W=${@:-cfg2 cfg3 cfg4 cfg5}
ROOTDIR=$(pwd)
export TMPDIR=/tmp
for w in $W; do
  OUT=$ROOTDIR/gpurun_out/$TAG/$w
  mkdir -p $OUT
  ARGS="--workload $w --no-configs --no-cpu-baseline --no-pcie --no-alongside --no-ct"
  python3 bench.py $ARGS > $OUT/bench.json 2> $OUT/bench.err || { tail -20 $OUT/bench.err; exit 1; }
  echo "$w: $(python3 -c "import json;l=json.load(open('$OUT/bench.json'));print(l['value'], l['unit'], l['ms_per_step'], 'ms/step')")"
  cd /tmp
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOTDIR/bench.py $ARGS --no-parity > $OUT/trace.log 2>&1 || { tail -20 $OUT/trace.log; exit 1; }
  for grp in "fetch:FETCH_SIZE" "write:WRITE_SIZE" "sq:SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY"; do
    name=${grp%%:*}; ctrs=${grp#*:}
    rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $OUT/pmc_$name -- python3 $ROOTDIR/bench.py $ARGS --no-parity --steps 10 --warmup 2 > $OUT/pmc_$name.log 2>&1 || { tail -20 $OUT/pmc_$name.log; exit 1; }
  done
  cd $ROOTDIR
done
echo profiled: $W
