#!/bin/bash
# Same-box A/B of two builds of the library on the bench workloads (GPU box):
#   tools/ab_bench.sh <libA.so> <libB.so> [workloads...]      (paths relative to the repo root)
# Box-to-box and run-to-run variation is ~2-3 %, so builds are only ever compared inside one call, alternating.
A=$PWD/$1; B=$PWD/$2; shift 2
W=${@:-cfg2 cfg3 cfg4 cfg5}
mkdir -p gpurun_out/ab
for rep in 1 2; do
  for w in $W; do
    FOURQ_AMD_LIB=$A python3 bench.py --workload $w --no-cpu-baseline > gpurun_out/ab/a_$w.json 2>/dev/null || exit 1
    FOURQ_AMD_LIB=$B python3 bench.py --workload $w --no-cpu-baseline > gpurun_out/ab/b_$w.json 2>/dev/null || exit 1
    python3 - $w <<PY
import json, sys
w = sys.argv[1]
a = json.load(open("gpurun_out/ab/a_%s.json" % w))["ms_per_step"]; b = json.load(open("gpurun_out/ab/b_%s.json" % w))["ms_per_step"]
print("%s  A %.4f ms   B %.4f ms   A is %+.1f%% faster" % (w, a, b, 100 * (b / a - 1)))
PY
  done
done
