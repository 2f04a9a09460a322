#!/bin/bash
# round 5, checkpoint: suite, profiles of every workload (rocprofv3 traces + PMC passes), their summary -- which writes
# profiles/pmc_traffic.json for THIS build of the library -- and only then the bench lines and the rehearsals of the N > 1 path, so that
# the committed lines carry roofline.traffic and valu_roofline.issue from counters of the same build.
# Afterwards, in the repo:  python tools/summarize_profiles.py gpurun_out/r05f r05  (profiles/ of the box is not merged back)
set -o pipefail
T=gpurun_out/r05f
mkdir -p $T
python -m pytest tests -m gpu -x -q > $T/pytest.log 2>&1 || { tail -40 $T/pytest.log; exit 1; }
tail -1 $T/pytest.log
bash tools/profile_all.sh r05f || exit 1
python3 tools/summarize_profiles.py $T r05 > $T/summary.txt 2>&1 || { tail -20 $T/summary.txt; exit 1; }
cat $T/summary.txt
python3 bench.py --steps 20 --warmup 5 > $T/bench_driver_args.json 2> $T/bench2.err || { tail -20 $T/bench2.err; exit 1; }
python3 bench.py > $T/bench.json 2> $T/bench.err || { tail -20 $T/bench.err; exit 1; }
for g in 2 4; do
  FOURQ_BENCH_REHEARSE=1 python3 bench.py --gpus $g --no-cpu-baseline > $T/rehearse_gpus$g.json 2> $T/rehearse$g.err || { tail -20 $T/rehearse$g.err; exit 1; }
done
python3 tools/single_call_probe.py > $T/single_call.txt 2>&1 || { tail -20 $T/single_call.txt; exit 1; }
python3 tools/perf_probe.py --modes endo_var,win_var,dh_endo,endo_fixed,win_fixed,dh_fixed,comb --sizes 16,18,20 > $T/perf_probe.txt 2>/dev/null
cat $T/perf_probe.txt
python3 -c "
import json
for f in ('bench_driver_args','bench','rehearse_gpus2','rehearse_gpus4'):
    l=json.load(open('$T/%s.json'%f)); print(f, l['value'], l['ms_per_step'], l['n_gpus'], l['config'].get('ranks_seen'), l['parity'].get('all_ranks_ok'), l.get('gather_ms'), l['roofline'].get('traffic'), l['valu_roofline']['issue'].get('frac'), l['clock']['in_kernel_mhz'], l['cycles_per_unit'])
"
