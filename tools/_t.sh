cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_gpu_batch.py tests/test_gpu_parity.py tests/test_gpu_loose_ends.py -x -q 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats -d gpurun_out/kt1 -o kt --output-format csv -- python bench.py --no-cpu-baseline --no-end-to-end --no-exclusive --repeats 1 --steps 10 --warmup 2 --groups 1 > gpurun_out/kt1.log 2>&1
python tools/kstats.py gpurun_out/kt1/kt_kernel_stats.csv 12 | grep -E "k_splice|k_copy|total"
rm -rf gpurun_out/kt1
python bench.py --no-cpu-baseline --no-end-to-end --no-exclusive 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('ms_per_step_regions'))"
