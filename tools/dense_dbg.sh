for m in ${MODES:-0 1 5 9 17 33 61}; do NP2_DENSE_DBG=$m python bench.py --no-cpu-baseline --no-end-to-end --workload ecoli --steps 10 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('dbg',$m,'ecoli ms',d['ms_per_step'],'diff ms',d['roofline']['avg_launch_ms'])
"; done
