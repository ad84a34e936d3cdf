#!/bin/bash
# Kernel-time table of one short bench run (run through gpurun from the repo root):
#   tools/quick_kt.sh <tag> [bench args...]  ->  gpurun_out/<tag>_kernels_per_step.txt (+ the bench line on stdout)
set -u
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out
timeout 300 python bench.py --no-cpu-baseline --no-end-to-end "$@" 2>/dev/null | tail -1 | cut -c1-260
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_kt -o kt --output-format csv -- python bench.py --no-cpu-baseline --no-end-to-end --no-exclusive --repeats 1 --steps 10 --warmup 2 "$@" > /dev/null 2>&1
python tools/kstats.py $OUT/${TAG}_kt/kt_kernel_stats.csv 12 > $OUT/${TAG}_kernels_per_step.txt
rm -rf $OUT/${TAG}_kt
