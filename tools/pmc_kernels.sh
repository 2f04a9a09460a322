#!/bin/bash
# rocprofv3 --pmc over a probe script, one pass per counter group, per-kernel means (GPU box):
#   tools/pmc_kernels.sh "<python script and args>" "<counters group 1>" ["<group 2>" ...]
export TMPDIR=/tmp
R=$PWD
PROBE=$1; shift
i=0
for group in "$@"; do
  i=$((i+1))
  out=$R/gpurun_out/pmck_$i
  rm -rf $out
  (cd /tmp && rocprofv3 --pmc $group --kernel-trace --output-format csv -d $out -- python3 $R/$PROBE > $out.log 2>&1) || { tail -5 $out.log; exit 1; }
  python3 - $(find $out -name "*counter_collection.csv" | head -1) <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in agg.items():
    if sum(len(x) for x in v.values()) >= 5 * len(v):
        name = k.split("(")[0].replace("void fq::(anonymous namespace)::", "")[:44]
        print("%-44s %s" % (name, "  ".join("%s=%d" % (c, round(sum(x) / len(x))) for c, x in sorted(v.items()))))
PY
done
