# Round 6, final evidence: counters of the final library (hash recorded), pmc_traffic.json rebuilt on the box from them, then the bench
# lines, the GPU suite and the fuzz run of the same library.  (profiles/ is not merged back: the rebuilt file is copied to gpurun_out/.)
set -x
mkdir -p gpurun_out
rm -rf gpurun_out/pmc_r06 gpurun_out/pmc_r06_1080p gpurun_out/pmc_r06_8k gpurun_out/prof_r06
bash tools/run_rocprof.sh r06 --steps 20 --warmup 5 > gpurun_out/rocprof_r06.log 2>&1
PMC_GROUPS="sq1 sq2 sq5 fetch write" bash tools/run_pmc.sh r06 > gpurun_out/pmc_r06.log 2>&1
PMC_GROUPS="sq1 fetch write" bash tools/run_pmc.sh r06_1080p --workload 1080p > gpurun_out/pmc_r06_1080p.log 2>&1
PMC_GROUPS="sq1 fetch write" bash tools/run_pmc.sh r06_8k --workload 8k > gpurun_out/pmc_r06_8k.log 2>&1
python tools/make_pmc_traffic.py gpurun_out/pmc_r06 4k 16 > /dev/null
python tools/make_pmc_traffic.py gpurun_out/pmc_r06_1080p 1080p 64 > /dev/null
python tools/make_pmc_traffic.py gpurun_out/pmc_r06_8k 8k 4 > /dev/null
cp profiles/pmc_traffic.json gpurun_out/pmc_traffic.json
bash profiles/r06_scripts/r06_session8_evidence.sh
