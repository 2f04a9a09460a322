#!/bin/bash
# Same-box A/B of two library builds on tools/perf_probe.py modes (GPU box):
#   tools/ab_probe.sh <libA.so> <libB.so> "<modes>" "<sizes>" [ENV=VAL ...]
A=$PWD/$1; B=$PWD/$2; MODES=$3; SIZES=$4; shift 4
for kv in "$@"; do export "$kv"; done
for rep in 1 2; do
  echo "--- A=$(basename $A) rep $rep"; FOURQ_AMD_LIB=$A python3 tools/perf_probe.py --modes $MODES --sizes $SIZES 2>/dev/null | grep "n="
  echo "--- B=$(basename $B) rep $rep"; FOURQ_AMD_LIB=$B python3 tools/perf_probe.py --modes $MODES --sizes $SIZES 2>/dev/null | grep "n="
done
