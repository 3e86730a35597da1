# Round 6, session 17: from-raw LoResDB window -- the counters (FETCH_SIZE / WRITE_SIZE / SQ_INSTS_VALU of the product), and the A/B at
# the other workloads and for one frame per call.
set -x
mkdir -p gpurun_out
PMC_GROUPS="sq1 fetch write" bash tools/run_pmc.sh r06b > gpurun_out/pmc_r06b.log 2>&1
python tools/pmc_summary.py gpurun_out/pmc_r06b | grep -i "final\|==" 
bash profiles/r06_scripts/r06_ab_variants.sh r06s17_1080p 2 --workload 1080p --pipeline --steps 100 --check -- lowbuf product
bash profiles/r06_scripts/r06_ab_variants.sh r06s17_8k 2 --workload 8k --pipeline --steps 100 --check -- lowbuf product
bash profiles/r06_scripts/r06_ab_variants.sh r06s17_4k_one_frame 2 --batch 1 --steps 400 --check -- lowbuf product
bash profiles/r06_scripts/r06_ab_variants.sh r06s17_4k_one_frame_pipelined 2 --batch 1 --pipeline --steps 400 --check -- lowbuf product
bash profiles/r06_scripts/r06_ab_variants.sh r06s17_1080p_one_frame 2 --workload 1080p --batch 1 --steps 400 --check -- lowbuf product
