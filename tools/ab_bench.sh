#!/bin/bash
# A/B of two builds of libnp2_hip.so on the default bench line, interleaved (run through gpurun from the repo root):
#   tools/ab_bench.sh <libA.so> <libB.so> [rounds] [bench args...]
A=$1; B=$2; R=${3:-3}; shift 3 || true
for i in $(seq 1 $R); do
  for L in $A $B; do
    v=$(NP2_LIB_PATH=$PWD/$L timeout 300 python bench.py --no-cpu-baseline --no-end-to-end --no-exclusive "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])")
    echo "$L $v"
  done
done
