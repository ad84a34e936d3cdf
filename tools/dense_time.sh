#!/bin/bash
# Dense-pass timing on one box: the E. coli-sized contig and the 60 Mb one (bench.py --workload ecoli [--scale 13]),
# kernel time from the HIP events around k_diff_reads.  usage (through gpurun): tools/dense_time.sh <tag>
TAG=${1:-dense}
for sc in 1 13; do
  python bench.py --workload ecoli --scale $sc --no-cpu-baseline --no-end-to-end --steps 10 --warmup 2 2> gpurun_out/${TAG}_s${sc}.err | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
r = d['roofline']
print('scale $sc: step %.3f ms, k_diff_reads %.4f ms, %.0f GB/s (frac %.4f)' % (d['ms_per_step'], r['avg_launch_ms'], r['achieved'], r['frac']))
"
done
