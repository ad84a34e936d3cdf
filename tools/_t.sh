cd "$GRAFT_REPO_ROOT"
for cpus in 0-1 0-3 0-7 0-15; do
LOCAL_WORLD_SIZE=8 taskset -c $cpus python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-end-to-end --no-exclusive 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('cpus $cpus (LOCAL_WORLD_SIZE=8: naps) value', d['value'], 'ms', d['ms_per_step'], d['ms_per_step_regions']['median'], 'host_cpu', d['host_cpu']['cpu_seconds_per_wall_second'], d['flush_ms']['per_group_totals_host_issue_wait'])"
done
for g in 1 2; do
LOCAL_WORLD_SIZE=8 taskset -c 0-1 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-end-to-end --no-exclusive --groups $g 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('cpus 0-1 groups $g value', d['value'], 'ms', d['ms_per_step'], d['ms_per_step_regions']['median'], 'host_cpu', d['host_cpu']['cpu_seconds_per_wall_second'])"
done
