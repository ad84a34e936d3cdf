#!/bin/bash
# SQ counters per kernel of one short bench run (run through gpurun from the repo root):
#   tools/quick_sq.sh <tag> [bench args...]  ->  gpurun_out/<tag>_pmc_sq.txt
set -u
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out
timeout 400 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAIT_INST_ANY --kernel-trace -d $OUT/${TAG}_ps -o s --output-format csv -- python bench.py --no-cpu-baseline --no-end-to-end --no-exclusive --repeats 1 --steps 3 --warmup 2 "$@" > /dev/null 2>&1
python tools/pmc_sq.py $OUT/${TAG}_ps/s_counter_collection.csv > $OUT/${TAG}_pmc_sq.txt
rm -rf $OUT/${TAG}_ps
