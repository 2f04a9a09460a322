#!/bin/bash
# the host-array pipeline at 2^20 elements, three I/O formats + cfg3's fixed-base call, against the shape of the pipeline:
# slots in flight, generations per inner chunk, who hands a slot on (GPU / host).  Then a kernel + copy timeline of the default shape.
set -e
OUT=gpurun_out/${1:-r05pipe}
mkdir -p $OUT
REPO=$PWD
export FOURQ_DEBUG_ROUTES=1
run() {  # label, [--timing] [--floor], env...
  label=$1; shift
  flags=""
  while [ "${1#--}" != "$1" ]; do flags="$flags $1"; shift; done
  echo "--- $label" >> $OUT/sweep.txt
  env "$@" python tools/pipeline_probe.py 20 --no-link --formats=r1,affine,bytes,fixed --reps=9 $flags 2>&1 | grep -v amdgpu.ids >> $OUT/sweep.txt
}
: > $OUT/sweep.txt
run "rounds 2-4 shape WITH their four timing events per chunk: 3 slots, 1 generation per chunk, host hands slots on" --timing FOURQ_PIPE_SLOTS=3 FOURQ_PIPE_GENS=1 FOURQ_PIPE_HOST_WAIT=1
run "the same without the timing events" FOURQ_PIPE_SLOTS=3 FOURQ_PIPE_GENS=1 FOURQ_PIPE_HOST_WAIT=1
run "4 slots, 1 gen, GPU hand-over" FOURQ_PIPE_SLOTS=4 FOURQ_PIPE_GENS=1
run "4 slots, 2 gens, GPU hand-over" FOURQ_PIPE_SLOTS=4 FOURQ_PIPE_GENS=2
run "DEFAULT: 4 slots, planned chunks, GPU hand-over (+ device-resident floors)" --floor FOURQ_PIPE_SLOTS=4
run "planned chunks, host hand-over" FOURQ_PIPE_HOST_WAIT=1
run "planned chunks, 3 slots" FOURQ_PIPE_SLOTS=3
run "planned chunks, 6 slots" FOURQ_PIPE_SLOTS=6
run "DEFAULT again" --floor FOURQ_PIPE_SLOTS=4
cat $OUT/sweep.txt
