cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_gpu_batch.py tests/test_gpu_frontend.py tests/test_ref_bundle.py tests/test_gpu_inflate.py tests/test_gpu_dist.py -x -q 2>&1 | tail -3
NP2_CLI_PROFILE=1 timeout 300 python tools/cli_probe.py > gpurun_out/cli_batch.log 2>&1
grep -E "^-t|batch driver|last record|contexts released" gpurun_out/cli_batch.log | head -40
for b in 8 16; do echo "NP2_CLI_BATCH=$b"; NP2_CLI_BATCH=$b timeout 300 python tools/cli_probe.py 2>&1 | grep "^-t"; done
python bench.py --no-cpu-baseline --no-end-to-end --no-exclusive 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('ms_per_step_regions'))"
