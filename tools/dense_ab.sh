#!/bin/bash
# Dense-pass experiments on one box (through gpurun): variants of libnp2_hip.so and of the generator's error rates on the
# 60 Mb contig, then the SQ counters of the same run.  usage: tools/dense_ab.sh <tag> [lib.so ...]
TAG=${1:-dab}; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
line() { python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('$1: step %.3f ms, k_diff_reads %.4f ms, %.0f GB/s (frac %.4f)' % (d['ms_per_step'], r['avg_launch_ms'], r['achieved'], r['frac']))"; }
B="python bench.py --workload ecoli --scale 13 --no-cpu-baseline --no-end-to-end --no-exclusive --repeats 1 --steps 10 --warmup 2"
for L in nextpolish2_amd/libnp2_hip.so "$@"; do
  for e in 1 0 2; do
    NP2_LIB_PATH=$PWD/$L NP2_BENCH_ERR_SCALE=$e timeout 300 $B 2>/dev/null | line "$L err x$e"
  done
done
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAIT_INST_ANY --kernel-trace -d gpurun_out/${TAG}_ps -o s --output-format csv -- $B --steps 3 > gpurun_out/${TAG}_ps.log 2>&1
python tools/pmc_sq.py gpurun_out/${TAG}_ps/s_counter_collection.csv > gpurun_out/${TAG}_pmc_sq.txt
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAVE_CYCLES SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU --kernel-trace -d gpurun_out/${TAG}_ps2 -o s --output-format csv -- $B --steps 3 > gpurun_out/${TAG}_ps2.log 2>&1
python tools/pmc_sq.py gpurun_out/${TAG}_ps2/s_counter_collection.csv > gpurun_out/${TAG}_pmc_sq2.txt
rm -rf gpurun_out/${TAG}_ps gpurun_out/${TAG}_ps2
head -8 gpurun_out/${TAG}_pmc_sq.txt; head -8 gpurun_out/${TAG}_pmc_sq2.txt
