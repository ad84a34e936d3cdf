cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace -d gpurun_out/kt1 -o kt --output-format csv -- python bench.py --no-cpu-baseline --no-end-to-end --no-exclusive --repeats 1 --steps 10 --warmup 2 --groups 1 > gpurun_out/gaps_kt1.log 2>&1
head -1 gpurun_out/kt1/kt_kernel_trace.csv
python tools/kcopies.py gpurun_out/kt1/kt_kernel_trace.csv 12
python tools/kgaps.py gpurun_out/kt1/kt_kernel_trace.csv 40 9
rm -rf gpurun_out/kt gpurun_out/kt1
