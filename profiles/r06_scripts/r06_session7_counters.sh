# Round 6, session 7: kernel trace + PMC passes of the LinearDepth-free library (device code hash recorded next to the counters).
set -x
mkdir -p gpurun_out
bash tools/run_rocprof.sh r06 --steps 20 --warmup 5 > gpurun_out/rocprof_r06.log 2>&1
PMC_GROUPS="sq1 sq2 sq5 fetch write" bash tools/run_pmc.sh r06 > gpurun_out/pmc_r06.log 2>&1
PMC_GROUPS="sq1 fetch write" bash tools/run_pmc.sh r06_1080p --workload 1080p > gpurun_out/pmc_r06_1080p.log 2>&1
PMC_GROUPS="sq1 fetch write" bash tools/run_pmc.sh r06_8k --workload 8k > gpurun_out/pmc_r06_8k.log 2>&1
cat gpurun_out/pmc_r06/code_sha256.txt
tail -5 gpurun_out/prof_r06/bench_under_rocprof.log | cut -c1-300
