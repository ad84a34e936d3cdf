cd $GRAFT_REPO_ROOT
R=r06
timeout 500 tools/collect_profiles.sh ${R}_yeast > /dev/null 2>&1
timeout 500 tools/collect_profiles.sh ${R}_yeast_one_group --groups 1 > /dev/null 2>&1
timeout 500 tools/collect_profiles.sh ${R}_ecoli --workload ecoli > /dev/null 2>&1
timeout 400 python bench.py > gpurun_out/${R}_yeast_bench_default_full.json 2> gpurun_out/${R}_yeast_bench_default_full.err
timeout 300 python bench.py --workload ecoli --no-cpu-baseline > gpurun_out/${R}_ecoli_bench_full.json 2> gpurun_out/${R}_ecoli_bench_full.err
timeout 300 python bench.py --workload ecoli --scale 13 --no-cpu-baseline --no-end-to-end --steps 10 --warmup 2 > gpurun_out/${R}_60Mb_contig_bench_line.json 2> gpurun_out/${R}_60Mb.err
NP2_BENCH_STAGES=1 timeout 600 python bench.py --scaling strong --workload chr1 --gpus 1 --steps 3 --warmup 1 > gpurun_out/${R}_strong_chr1_1gpu.json 2> gpurun_out/${R}_strong_chr1_1gpu.err
timeout 600 python bench.py --scaling strong --workload chr1 --gpus 1 --steps 3 --warmup 1 --haploid > gpurun_out/${R}_strong_chr1_1gpu_haploid.json 2> gpurun_out/${R}_strong_chr1_1gpu_haploid.err
NP2_PHASE_PROFILE=1 timeout 600 python bench.py --scaling strong --workload chr1 --gpus 1 --steps 2 --warmup 1 2>&1 > /dev/null | grep -E "vote host|losing_reads|sweeps:|pieces:|aggregate|local_moving" | tail -40 > gpurun_out/${R}_strong_chr1_vote_phases.txt
NP2_IO_PROFILE=1 timeout 300 python tools/bench_frontend.py 4641652 > gpurun_out/${R}_frontend_ecoli_size.log 2>&1
timeout 600 python tools/inflate_probe.py > gpurun_out/${R}_device_read_extraction_ecoli_size.log 2>&1
NP2_INF_PROF=1 timeout 300 python tools/inflate_only.py 20000 1200000 4641652 > gpurun_out/${R}_inflate_kernel_sizes.log 2>&1
for p in 1 2 4; do echo "NP2_INF_PROBE=$p"; NP2_INF_PROBE=$p timeout 300 python tools/inflate_only.py 4641652 2>&1 | grep "^L"; done > gpurun_out/${R}_inflate_kernel_probe.log 2>&1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --kernel-trace -d gpurun_out/inf_ps -o s --output-format csv -- python tools/inflate_only.py 4641652 > /dev/null 2>&1
python tools/pmc_sq.py gpurun_out/inf_ps/s_counter_collection.csv | grep -i "inflate\|^kernel" > gpurun_out/${R}_inflate_pmc_sq.txt; rm -rf gpurun_out/inf_ps
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d gpurun_out/inf_pf -o f --output-format csv -- python tools/inflate_only.py 4641652 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d gpurun_out/inf_pw -o w --output-format csv -- python tools/inflate_only.py 4641652 > /dev/null 2>&1
python tools/pmc_summary.py gpurun_out/inf_pf/f_counter_collection.csv gpurun_out/inf_pw/w_counter_collection.csv > gpurun_out/${R}_inflate_pmc_fetch_write.json 2>&1; rm -rf gpurun_out/inf_pf gpurun_out/inf_pw
NP2_CLI_PROFILE=1 timeout 300 python tools/cli_probe.py > gpurun_out/${R}_cli_assembly_probe.log 2>&1
python tools/pf_prof.py > gpurun_out/${R}_pf_tile_phases.txt 2>&1
timeout 600 python tools/cli_split_probe.py > gpurun_out/${R}_cli_split_probe.log 2>&1
for c in 0-1 0-3 0-7 0-15; do LOCAL_WORLD_SIZE=8 taskset -c $c python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-end-to-end --no-exclusive 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('taskset -c $c, LOCAL_WORLD_SIZE=8 (waits nap):', d['value'], 'Mbp/s,', d['ms_per_step'], 'ms per step (median', d['ms_per_step_regions']['median'], '), host CPUs busy', d['host_cpu']['cpu_seconds_per_wall_second'])"; done > gpurun_out/${R}_rank_cpu_budget.txt 2>&1
for g in 4 5 6 8; do python bench.py --no-cpu-baseline --no-end-to-end --no-exclusive --groups $g 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('--groups $g:', d['value'], 'Mbp/s,', d['ms_per_step'], 'ms per step; per-group call ms', d['flush_ms']['call_breakdown_ms_per_group']['caller_clock'])"; done > gpurun_out/${R}_group_counts.txt 2>&1
timeout 300 python tools/vote_small_probe.py 2>&1 | tail -18 > gpurun_out/${R}_vote_host_yeast_chromosome.txt
timeout 300 python tools/cli_rss_probe.py > gpurun_out/${R}_cli_fresh_process.txt 2>&1
ls gpurun_out | grep "^${R}_" | head -80
