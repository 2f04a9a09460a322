#!/bin/bash
# round 4, checkpoint: suite, profiles of every workload (rocprofv3 traces + PMC passes), their summary -- which writes
# profiles/pmc_traffic.json for THIS build of the library -- and only then the default bench lines and the rehearsals of the
# N > 1 path, so that the committed lines carry roofline.traffic and valu_roofline.issue from counters of the same build.
# Afterwards, in the repo:  python tools/summarize_profiles.py gpurun_out/r04f r04  (profiles/ of the box is not merged back)
set -o pipefail
mkdir -p gpurun_out/r04f
python -m pytest tests -m gpu -x -q > gpurun_out/r04f/pytest.log 2>&1 || { tail -40 gpurun_out/r04f/pytest.log; exit 1; }
tail -1 gpurun_out/r04f/pytest.log
bash tools/profile_all.sh r04f || exit 1
python3 tools/summarize_profiles.py gpurun_out/r04f r04 > gpurun_out/r04f/summary.txt 2>&1 || { tail -20 gpurun_out/r04f/summary.txt; exit 1; }
cat gpurun_out/r04f/summary.txt
python3 bench.py > gpurun_out/r04f/bench.json 2> gpurun_out/r04f/bench.err || { tail -20 gpurun_out/r04f/bench.err; exit 1; }
python3 bench.py --steps 20 --warmup 5 > gpurun_out/r04f/bench_driver_args.json 2> gpurun_out/r04f/bench2.err || exit 1
for g in 2 4; do
  FOURQ_BENCH_REHEARSE=1 python3 bench.py --gpus $g --no-cpu-baseline > gpurun_out/r04f/rehearse_gpus$g.json 2> gpurun_out/r04f/rehearse$g.err || { tail -20 gpurun_out/r04f/rehearse$g.err; exit 1; }
done
python3 tools/single_call_probe.py > gpurun_out/r04f/single_call.txt 2>&1 || { tail -20 gpurun_out/r04f/single_call.txt; exit 1; }
cat gpurun_out/r04f/single_call.txt
python3 tools/quad_probe.py > gpurun_out/r04f/quad_probe.txt 2> gpurun_out/r04f/quad_probe.err || { tail -20 gpurun_out/r04f/quad_probe.err; exit 1; }
python3 tools/perf_probe.py --modes endo_var,win_var,dh_endo,endo_fixed,win_fixed,dh_fixed,comb --sizes 16,18,20 > gpurun_out/r04f/perf_probe.txt 2>/dev/null
cat gpurun_out/r04f/perf_probe.txt
python3 -c "
import json
for f in ('bench','bench_driver_args','rehearse_gpus2','rehearse_gpus4'):
    l=json.load(open('gpurun_out/r04f/%s.json'%f)); print(f, l['value'], l['ms_per_step'], l['n_gpus'], l['parity'].get('all_ranks_ok'), l.get('gather_ms'), l['roofline'].get('traffic'), l['valu_roofline']['issue'].get('frac'))
"
