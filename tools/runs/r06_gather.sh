#!/bin/bash
# VERDICT r5 item 5, ONE experiment: N, D of the fused kernels' per-lane slots as four 128-bit canonical words (64 B per entry, FQ_ND_PACKED=1)
# against the dense limb layout (80 B): same-box kernel time (bench, alternating) and the two counters that decide -- FETCH_SIZE and
# SQ_INSTS_VALU of the headline kernel -- one --pmc pass each (never combined with other trace domains).
#   python -m fourq_amd.build --out variants/libfourq_ndpacked.so -DFQ_ND_PACKED=1      (build box)
#   tools/runs/r06_gather.sh > gpurun_out/r06_gather.txt
export TMPDIR=/tmp
R=$PWD
ARGS="--workload cfg2 --no-configs --no-cpu-baseline --no-pcie --no-alongside --no-ct"
echo "# same-box A/B, cfg2 (A = shipped dense limbs, B = packed words), ms per step of 2^16 MUL_endo"
tools/ab_bench.sh fourq_amd/libfourq_amd.so variants/libfourq_ndpacked.so cfg2 cfg5 || exit 1
for lib in fourq_amd/libfourq_amd.so variants/libfourq_ndpacked.so; do
  for ctr in FETCH_SIZE WRITE_SIZE SQ_INSTS_VALU; do
    out=$R/gpurun_out/r06_gather_pmc; rm -rf $out
    (cd /tmp && FOURQ_AMD_LIB=$R/$lib rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $out -- python3 $R/bench.py $ARGS --no-parity --steps 10 --warmup 2 > $out.log 2>&1) || { tail -5 $out.log; exit 1; }
    python3 - $(find $out -name "*counter_collection.csv" | head -1) $lib $ctr <<'PY'
import csv, sys, collections
vals = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if "ladder_kernel" in r["Kernel_Name"] and r["Counter_Name"] == sys.argv[3]:
        vals[r["Kernel_Name"].split("(")[0][-44:]].append(float(r["Counter_Value"]))
for k, v in vals.items():
    if len(v) >= 10:
        m = sum(v) / len(v)
        # FETCH_SIZE / WRITE_SIZE are KiB; on gfx950 FETCH_SIZE counts a 128-byte request at 64 bytes for 16-byte-per-lane reads: doubled (MI355X_MICROARCH.md)
        extra = {"FETCH_SIZE": "  = %.1f MB per launch after the gfx950 correction (x 2)" % (2 * m * 1024 / 1e6), "WRITE_SIZE": "  = %.1f MB per launch" % (m * 1024 / 1e6),
                 "SQ_INSTS_VALU": "  = %.0f per lane (1 024 waves)" % (m / 1024)}[sys.argv[3]]
        print("%-34s %-14s %s mean %.1f over %d launches%s" % (sys.argv[2], sys.argv[3], k, m, len(v), extra))
PY
  done
done
