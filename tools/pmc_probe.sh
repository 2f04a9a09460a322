#!/bin/bash
# tools/pmc_probe.sh <lib> <mode> <lgsize> : PMC counters for one perf_probe configuration (GPU box)
set -o pipefail
LIB=$1; MODE=$2; LG=$3
ROOTDIR=$(pwd)
OUT=$ROOTDIR/gpurun_out/pmc_$(basename $LIB .so)_${MODE}_$LG
mkdir -p $OUT
export TMPDIR=/tmp FOURQ_AMD_LIB=$ROOTDIR/$LIB
cd /tmp
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $OUT -- python3 $ROOTDIR/tools/perf_probe.py --sizes $LG --modes $MODE --reps 3 > $OUT/log.txt 2>&1 || { tail -5 $OUT/log.txt; exit 1; }
cd $ROOTDIR
python3 - <<PY
import csv, glob, collections
f = glob.glob("$OUT/*/*_counter_collection.csv")[0]
kt = glob.glob("$OUT/*/*_kernel_trace.csv")[0]
dur = collections.defaultdict(list)
for r in csv.DictReader(open(kt)):
    dur[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in agg.items():
    if "ladder" not in k: continue
    d = sorted(dur[k])[len(dur[k]) // 2]
    m = {n: sorted(v)[len(v) // 2] for n, v in c.items()}
    print("$LIB $MODE 2^$LG", k[:60], "dur_us=%.1f" % (d / 1e3))
    print("   ", {n: "%.4g" % v for n, v in m.items()})
    if "GRBM_GUI_ACTIVE" in m: print("    eff clock MHz ~ %.0f" % (m["GRBM_GUI_ACTIVE"] / 8 / (d / 1e3)))
    if "SQ_INSTS_VALU" in m and "SQ_BUSY_CYCLES" in m:
        print("    VALU instr/wave = %.0f ; busy cycles(sum)=%.4g" % (m["SQ_INSTS_VALU"] / m["SQ_WAVES"], m["SQ_BUSY_CYCLES"]))
PY
