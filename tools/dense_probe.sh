#!/bin/bash
# Where the dense pass's time goes (through gpurun): parts of k_diff_reads launched once more after the real pass on the
# 60 Mb contig (NP2_DENSE_PROBE: 1 = loads + counts, 2 = + carries, contig windows, compare, 3 = + checkpoint stores,
# 4 = + dirty queue = all of phase 1, 5 = + phase 2 up to the bucket reservation).  usage: tools/dense_probe.sh [bench args]
for m in 1 2 3 4 5; do
  echo -n "probe $m: "
  NP2_DENSE_PROBE=$m timeout 300 python bench.py --workload ecoli --scale 13 --no-cpu-baseline --no-end-to-end --no-exclusive --repeats 1 --steps 6 --warmup 2 "$@" 2>&1 >/dev/null | grep diff_probe | sort | head -3 | tail -1
done
