cd $GRAFT_REPO_ROOT
timeout 500 tools/collect_profiles.sh r05_yeast > /dev/null 2>&1
timeout 500 tools/collect_profiles.sh r05_yeast_one_group --groups 1 > /dev/null 2>&1
timeout 500 tools/collect_profiles.sh r05_ecoli --workload ecoli > /dev/null 2>&1
timeout 400 python bench.py > gpurun_out/r05_yeast_bench_default_full.json 2> gpurun_out/r05_yeast_bench_default_full.err
timeout 300 python bench.py --workload ecoli --no-cpu-baseline > gpurun_out/r05_ecoli_bench_full.json 2> gpurun_out/r05_ecoli_bench_full.err
timeout 300 python bench.py --workload ecoli --scale 13 --no-cpu-baseline --no-end-to-end --steps 10 --warmup 2 > gpurun_out/r05_60Mb_contig_bench_line.json 2> gpurun_out/r05_60Mb.err
timeout 600 python bench.py --scaling strong --workload chr1 --gpus 1 --steps 3 --warmup 1 > gpurun_out/r05_strong_chr1_1gpu.json 2> gpurun_out/r05_strong_chr1_1gpu.err
NP2_IO_PROFILE=1 timeout 300 python tools/bench_frontend.py 4641652 > gpurun_out/r05_frontend_ecoli_size.log 2>&1
NP2_CLI_PROFILE=1 timeout 300 python tools/cli_probe.py > gpurun_out/r05_cli_assembly_probe.log 2>&1
ls gpurun_out | grep "^r05_" | head -80
