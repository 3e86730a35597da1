# usage: bash tools/collect_round.sh <tag> [frames-per-launch=16]   (here, after `gpurun ... tools/run_gpu_round.sh <tag>`)
# copies the judged evidence of a round from gpurun_out/ (scratch) into profiles/ (tracked)
set -e
T=$1; N=${2:-16}
python tools/rocprof_timed_region.py gpurun_out/prof_$T/trace_kernel_trace.csv 30 > profiles/${T}_kernel_trace_timed_region.txt
cp gpurun_out/prof_$T/trace_kernel_stats.csv profiles/${T}_kernel_stats.csv
cp gpurun_out/bench_$T.json profiles/${T}_bench_4k_batch16.json
cp gpurun_out/bench_${T}_1080p.json profiles/${T}_bench_1080p_batch64.json
cp gpurun_out/bench_${T}_8k.json profiles/${T}_bench_8k_f16_batch4.json
cp gpurun_out/bench_${T}_shaded.json profiles/${T}_bench_4k_shaded.json
cp gpurun_out/bench_force_dist_$T.log profiles/${T}_bench_force_dist.log
cp gpurun_out/fuzz_$T.log profiles/${T}_fuzz_gpu.log
cp gpurun_out/pytest_gpu_$T.log profiles/${T}_pytest_gpu.log
cp gpurun_out/ubench_issue_$T.txt profiles/${T}_ubench_issue.txt
cp gpurun_out/ubench_lds_$T.txt profiles/${T}_ubench_lds.txt
[ -f gpurun_out/bench_${T}_two_ranks_one_frame_each_gloo.log ] && grep '^{' gpurun_out/bench_${T}_two_ranks_one_frame_each_gloo.log > profiles/${T}_bench_4k_two_ranks_one_frame_each_gloo.json
[ -f gpurun_out/pool_enqueue_cost_$T.jsonl ] && cp gpurun_out/pool_enqueue_cost_$T.jsonl profiles/${T}_pool_enqueue_cost_worker_threads.jsonl
for f in pool2 pool3; do [ -f gpurun_out/bench_${T}_$f.log ] && grep '^{' gpurun_out/bench_${T}_$f.log > profiles/${T}_bench_4k_$f.json; done
for f in plain pipelined; do [ -f gpurun_out/phase_clocks_${f}_$T.json ] && cp gpurun_out/phase_clocks_${f}_$T.json profiles/${T}_phase_clocks_$f.json; done
[ -f gpurun_out/ubench_launch_$T.txt ] && cp gpurun_out/ubench_launch_$T.txt profiles/${T}_ubench_launch.txt
python tools/pmc_summary.py gpurun_out/pmc_$T > profiles/${T}_pmc_summary.txt
python tools/make_pmc_traffic.py gpurun_out/pmc_$T 4k $N > /dev/null
ls -la profiles/${T}_*
