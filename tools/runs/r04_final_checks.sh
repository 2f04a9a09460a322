#!/bin/bash
# round 3, final build: differential fuzz against the C oracle (random sizes around every route boundary, random knobs) and the 2^22 soak, both selection modes
set -o pipefail
mkdir -p gpurun_out/r04f
python3 tools/fuzz.py 240 > gpurun_out/r04f/fuzz.txt 2>&1 || { tail -30 gpurun_out/r04f/fuzz.txt; exit 1; }
tail -25 gpurun_out/r04f/fuzz.txt
python3 tools/soak.py > gpurun_out/r04f/soak.txt 2>&1 || { tail -30 gpurun_out/r04f/soak.txt; exit 1; }
FOURQ_CT_SELECT=1 python3 tools/soak.py > gpurun_out/r04f/soak_ct.txt 2>&1 || { tail -30 gpurun_out/r04f/soak_ct.txt; exit 1; }
tail -n 3 gpurun_out/r04f/soak.txt; tail -n 3 gpurun_out/r04f/soak_ct.txt
