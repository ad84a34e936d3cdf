#!/bin/bash
# Collects the rocprofv3 evidence of one round on a GPU box (run through gpurun from the repo root):
#   tools/collect_profiles.sh <tag> [bench args...]   ->  gpurun_out/<tag>_*  (copy what you want judged into profiles/)
# Counter passes are separate runs with --kernel-trace only (never combined with the sys / hip / hsa trace domains).
set -u
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out
ARGS="--no-cpu-baseline --no-end-to-end --no-exclusive --repeats 1 --steps 10 --warmup 2 $*"
python bench.py --no-cpu-baseline --no-end-to-end "$@" > $OUT/${TAG}_bench_line.json 2> $OUT/${TAG}_bench.err
rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_kt -o kt --output-format csv -- python bench.py $ARGS > $OUT/${TAG}_kt.log 2>&1
cp $OUT/${TAG}_kt/kt_kernel_stats.csv $OUT/${TAG}_bench_kernel_stats.csv
python tools/kstats.py $OUT/${TAG}_bench_kernel_stats.csv 12 > $OUT/${TAG}_kernels_per_step.txt
python tools/kbusy.py $OUT/${TAG}_kt/kt_kernel_trace.csv 50 > $OUT/${TAG}_gpu_busy.txt
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/${TAG}_pf -o f --output-format csv -- python bench.py $ARGS --steps 3 > $OUT/${TAG}_pf.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/${TAG}_pw -o w --output-format csv -- python bench.py $ARGS --steps 3 > $OUT/${TAG}_pw.log 2>&1
python tools/pmc_summary.py $OUT/${TAG}_pf/f_counter_collection.csv $OUT/${TAG}_pw/w_counter_collection.csv > $OUT/${TAG}_pmc_fetch_write.json
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAIT_INST_ANY --kernel-trace -d $OUT/${TAG}_ps -o s --output-format csv -- python bench.py $ARGS --steps 3 > $OUT/${TAG}_ps.log 2>&1
python tools/pmc_sq.py $OUT/${TAG}_ps/s_counter_collection.csv > $OUT/${TAG}_pmc_sq.txt
rm -rf $OUT/${TAG}_kt $OUT/${TAG}_pf $OUT/${TAG}_pw $OUT/${TAG}_ps
ls -la $OUT/${TAG}_*
