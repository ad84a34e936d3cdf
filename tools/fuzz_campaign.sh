#!/bin/bash
# A fuzz campaign on a GPU box: HIP path vs oracle over seeded random cases (tests/tools/fuzz_*.py), several streams side
# by side (the oracle is the slow, single-threaded part).  Logs under gpurun_out/fuzz/; the last line of each says "bad N".
#   bash tools/fuzz_campaign.sh [light cases per stream] [wide cases per stream] [heavy cases per stream] [seed offset]
# (the seed offset moves every stream to fresh seeds: a second campaign covers new cases instead of repeating the first)
export TMPDIR=/tmp
NL=${1:-2000}; NW=${2:-2000}; NH=${3:-60}; SO=${4:-0}
mkdir -p gpurun_out/fuzz
wait_all() { for i in $(seq 1 400); do n=$(pgrep -c -f "^python tests/tools/fuzz_" || true); [ "$n" = "0" ] && break; sleep 3; done; }
for s in 5001 5002 5003 5004; do s=$((s+SO)); (timeout 1500 python tests/tools/fuzz_polish.py $s $NL > gpurun_out/fuzz/polish_$s.log 2>&1 &) ; done
for s in 6001 6002 6003 6004 6005 6006; do s=$((s+SO)); (timeout 1500 python tests/tools/fuzz_polish.py $s $NW wide > gpurun_out/fuzz/wide_$s.log 2>&1 &) ; done
for s in 4003 4004; do s=$((s+SO)); (timeout 1500 python tests/tools/fuzz_front.py $s 400 > gpurun_out/fuzz/front_$s.log 2>&1 &) ; done
sleep 5; wait_all
for s in 3002 3003; do s=$((s+SO)); timeout 900 python tests/tools/fuzz_polish.py $s $NH heavy > gpurun_out/fuzz/polish_heavy_$s.log 2>&1; done
timeout 600 python tests/tools/fuzz_reuse.py $((78+SO)) 20 > gpurun_out/fuzz/reuse_$((78+SO)).log 2>&1
for f in gpurun_out/fuzz/*.log; do echo "$f: $(tail -n 1 $f)"; done
grep -h "MISMATCH" gpurun_out/fuzz/*.log | head -20
