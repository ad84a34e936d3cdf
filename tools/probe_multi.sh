#!/bin/bash
# run P simultaneous probe processes with NCTX contexts each
P=$1; N=$2
for i in $(seq 1 $P); do NCTX=$N python tools/diploid_probe.py > gpurun_out/pm_${P}_${N}_$i.log 2>&1 & done
wait
for i in $(seq 1 $P); do echo "P=$P N=$N proc $i: $(grep 'rep 2' gpurun_out/pm_${P}_${N}_$i.log)"; done
