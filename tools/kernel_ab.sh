#!/bin/bash
# Per-kernel durations (rocprofv3 --kernel-trace --stats) of tools/perf_probe.py modes for one or more library builds (GPU box):
#   tools/kernel_ab.sh "<modes>" "<sizes>" <lib.so> [<lib.so> ...] [-- ENV=VAL ...]
MODES=$1; SIZES=$2; shift 2
LIBS=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do LIBS+=("$1"); shift; done
[ "$1" == "--" ] && shift
for kv in "$@"; do export "$kv"; done
ROOTDIR=$(pwd); export TMPDIR=/tmp
for lib in "${LIBS[@]}"; do
  name=$(basename $lib .so); out=$ROOTDIR/gpurun_out/kab_$name; rm -rf $out
  export FOURQ_AMD_LIB=$ROOTDIR/$lib
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 $ROOTDIR/tools/perf_probe.py --modes $MODES --sizes $SIZES --reps 10 > $out.log 2>&1) || { tail -5 $out.log; exit 1; }
  echo "=== $name"; python3 - $out <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/*/*_kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        n = r["Name"].replace("void fq::(anonymous namespace)::", "").replace("(anonymous namespace)::", "").split("(")[0]
        if int(r["Calls"]) >= 5 and ("ladder" in n or "prep" in n or "comb" in n or "normalize" in n):
            print("  %-52s calls %4s  avg %9.1f us  min %9.1f us" % (n, r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3))
PY
done
