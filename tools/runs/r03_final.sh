#!/bin/bash
# round 3, checkpoint: suite, the default bench line, rehearsals of the N > 1 path, profiles of every workload
set -o pipefail
mkdir -p gpurun_out/r03f
python -m pytest tests -m gpu -x -q > gpurun_out/r03f/pytest.log 2>&1 || { tail -40 gpurun_out/r03f/pytest.log; exit 1; }
tail -1 gpurun_out/r03f/pytest.log
python3 bench.py > gpurun_out/r03f/bench.json 2> gpurun_out/r03f/bench.err || { tail -20 gpurun_out/r03f/bench.err; exit 1; }
python3 bench.py --steps 20 --warmup 5 > gpurun_out/r03f/bench_driver_args.json 2> gpurun_out/r03f/bench2.err || exit 1
for g in 2 4; do
  FOURQ_BENCH_REHEARSE=1 python3 bench.py --gpus $g --no-cpu-baseline > gpurun_out/r03f/rehearse_gpus$g.json 2> gpurun_out/r03f/rehearse$g.err || { tail -20 gpurun_out/r03f/rehearse$g.err; exit 1; }
done
python3 -c "
import json
for f in ('bench','bench_driver_args','rehearse_gpus2','rehearse_gpus4'):
    l=json.load(open('gpurun_out/r03f/%s.json'%f)); print(f, l['value'], l['ms_per_step'], l['n_gpus'], l['parity'].get('all_ranks_ok'), l.get('gather_ms'))
"
bash tools/profile_all.sh r03f
