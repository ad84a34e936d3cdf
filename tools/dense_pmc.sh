#!/bin/bash
# PMC instruction counts of k_diff_reads on the E. coli-sized contig (run through gpurun): tools/dense_pmc.sh <tag>
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out
for m in ${MODES:-0 1}; do
NP2_DENSE_DBG=$m rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace -d $OUT/$1_ps$m -o s --output-format csv -- python bench.py --no-cpu-baseline --no-end-to-end --no-exclusive --workload ecoli --steps 3 --warmup 1 > /dev/null 2>&1
python tools/pmc_sq.py $OUT/$1_ps$m/s_counter_collection.csv | grep -E "^kernel|k_diff_reads" > $OUT/$1_pmc_dbg$m.txt
rm -rf $OUT/$1_ps$m
done
cat $OUT/$1_pmc_dbg*.txt
