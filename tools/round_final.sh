#!/bin/bash
# One round's closing measurements on the GPU box (gpurun), by stage; replaces tools/runs/rNN_final.sh, rNN_final_checks.sh and
# rNN_ct_evidence.sh of rounds 3-5.  Everything lands under gpurun_out/<tag>f/ (merged back by gpurun).
#   tools/round_final.sh <tag> suite      GPU test suite
#   tools/round_final.sh <tag> profiles   per workload: bench line, rocprofv3 kernel trace + stats, three PMC passes (tools/profile_all.sh), their summary
#                                         -- which writes profiles/pmc_traffic.json FOR THIS BUILD on the box, so that the bench stage's lines carry
#                                         roofline.traffic and valu_roofline.issue from counters of the same build
#   tools/round_final.sh <tag> bench      the driver's line (--steps 20 --warmup 5, three runs), the default line, the N > 1 rehearsals, probes
#   tools/round_final.sh <tag> checks     differential fuzz against the C oracle + the 2^22 soak, both selection modes
#   tools/round_final.sh <tag> ct         counter invariance of the constant-time kernels (tools/ct_invariance.sh)
# Afterwards, in the repo:  python tools/summarize_profiles.py gpurun_out/<tag>f <tag>     (profiles/ of the box is not merged back)
set -o pipefail
TAG=${1:?tag, e.g. r06}; shift
STAGES=${@:-suite profiles bench}
T=gpurun_out/${TAG}f
mkdir -p $T
for stage in $STAGES; do
  case $stage in
  suite)
    python -m pytest tests -m gpu -x -q > $T/pytest.log 2>&1 || { tail -40 $T/pytest.log; exit 1; }
    tail -1 $T/pytest.log ;;
  profiles)
    bash tools/profile_all.sh ${TAG}f || exit 1
    python3 tools/summarize_profiles.py $T $TAG > $T/summary.txt 2>&1 || { tail -20 $T/summary.txt; exit 1; }
    cat $T/summary.txt ;;
  bench)
    for k in 1 2 3; do
      python3 bench.py --steps 20 --warmup 5 --full-json $T/bench_driver_args_run$k.full.json > $T/bench_driver_args_run$k.json 2> $T/bench_d$k.err || { tail -20 $T/bench_d$k.err; exit 1; }
    done
    python3 bench.py --full-json $T/bench.full.json > $T/bench.json 2> $T/bench.err || { tail -20 $T/bench.err; exit 1; }
    for w in cfg3 cfg4 cfg5; do
      python3 bench.py --workload $w --no-configs --full-json $T/bench_$w.full.json > $T/bench_$w.json 2> $T/bench_$w.err || { tail -20 $T/bench_$w.err; exit 1; }
    done
    for g in 2 4; do
      FOURQ_BENCH_REHEARSE=1 python3 bench.py --gpus $g --full-json $T/rehearse_gpus$g.full.json > $T/rehearse_gpus$g.json 2> $T/rehearse$g.err || { tail -20 $T/rehearse$g.err; exit 1; }
    done
    python3 tools/single_call_probe.py > $T/single_call.txt 2>&1 || { tail -20 $T/single_call.txt; exit 1; }
    python3 tools/perf_probe.py --modes endo_var,win_var,dh_endo,endo_fixed,win_fixed,dh_fixed,comb --sizes 16,18,20 > $T/perf_probe.txt 2>/dev/null
    python3 - $T <<'PY'
import json, sys
T = sys.argv[1]
for f in ("bench_driver_args_run1", "bench_driver_args_run2", "bench_driver_args_run3", "bench", "bench_cfg3", "bench_cfg4", "bench_cfg5", "rehearse_gpus2", "rehearse_gpus4"):
    raw = open("%s/%s.json" % (T, f)).read()
    l = json.loads(raw)
    print("%-24s %5d bytes  value %.4g  ms %.4f  n_gpus %d  ranks %s  parity %s  gather %s  traffic %s  issue %s  %s MHz  %s cycles/unit" % (
        f, len(raw), l["value"], l["ms_per_step"], l["n_gpus"], l["config"]["ranks_seen"], l["parity_ok"], l.get("gather_ms"), l["roofline"]["traffic"],
        l["valu_roofline"]["issue_frac"], l["clock_mhz"], l["cycles_per_unit"]))
PY
    ;;
  checks)
    python3 tools/fuzz.py ${FUZZ_SECONDS:-240} > $T/fuzz.txt 2>&1 || { tail -30 $T/fuzz.txt; exit 1; }
    tail -25 $T/fuzz.txt
    python3 tools/soak.py > $T/soak.txt 2>&1 || { tail -30 $T/soak.txt; exit 1; }
    FOURQ_CT_SELECT=1 python3 tools/soak.py > $T/soak_ct.txt 2>&1 || { tail -30 $T/soak_ct.txt; exit 1; }
    tail -n 3 $T/soak.txt; tail -n 3 $T/soak_ct.txt ;;
  ct)
    mkdir -p $T/ct
    bash tools/ct_invariance.sh 65536 > $T/ct/full.txt 2>&1 || { tail -20 $T/ct/full.txt; exit 1; }
    CT_CLASSES="same ones zero random" CT_TRACE_ONLY=1 bash tools/ct_invariance.sh 65536 > $T/ct/reversed.txt 2>&1 || { tail -20 $T/ct/reversed.txt; exit 1; }
    CT_MODES=ct bash tools/ct_invariance.sh 4096 > $T/ct/quad.txt 2>&1 || { tail -20 $T/ct/quad.txt; exit 1; }
    CT_MODES=ct bash tools/ct_invariance.sh 20000 > $T/ct/pair.txt 2>&1 || { tail -20 $T/ct/pair.txt; exit 1; }
    grep -c "0.00 %" $T/ct/full.txt $T/ct/quad.txt $T/ct/pair.txt ;;
  *) echo "unknown stage $stage"; exit 2 ;;
  esac
done
