#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r03
python -m pytest tests -m gpu -x -q > gpurun_out/r03/pytest4.log 2>&1 || { tail -40 gpurun_out/r03/pytest4.log; exit 1; }
tail -2 gpurun_out/r03/pytest4.log
S=$(date +%s); python3 bench.py > gpurun_out/r03/bench_default.json 2> gpurun_out/r03/bench_default.err || { tail -20 gpurun_out/r03/bench_default.err; exit 1; }; echo "Elapsed $(( $(date +%s) - S )) s" >> gpurun_out/r03/bench_default.err
grep "Elapsed" gpurun_out/r03/bench_default.err
python3 - <<'PY'
import json
l=json.load(open("gpurun_out/r03/bench_default.json"))
print(l["value"], l["ms_per_step"], {k:(v["ms_per_step"], v["ratio_vs_default"]) for k,v in l["ct_select"].items() if k!="mode"})
for k,v in l["configs"].items(): print(k, v["ms_per_step"], v["value"], v["pcie_inclusive"]["value"], v["valu_roofline"]["executed_frac_of_measured_peak"])
print(l["roofline"]["traffic"], l["roofline"]["traffic_source"].get("note"))
print(l["pcie_inclusive"]["value"], l["cpu_baseline"]["value"])
PY
