# Round 6, session 24: the fused last kernel with its LoResDB window written behind the V-blur (17.6 KB of LDS, eight workgroups per CU) --
# parity (everything that reaches the fused kernel), then product vs variant `wg7` (22.8 KB, seven), alternating.
set -x
mkdir -p gpurun_out
python tools/parity_probe.py 2>&1 | grep BAD | cut -c1-300
timeout 1200 python -m pytest tests/test_from_raw_window.py tests/test_large_pipelined.py tests/test_reference_goldens.py tests/test_hostile_depth.py tests/test_pipelined.py tests/test_variants_gpu.py -m gpu -q -x 2>&1 | tail -3
timeout 600 python tools/fuzz_gpu.py 3000 1200000 2>&1 | tail -1
bash profiles/r06_scripts/r06_ab_variants.sh r06s24_pipelined 4 --pipeline --steps 100 --check -- wg7 product
bash profiles/r06_scripts/r06_ab_variants.sh r06s24_1080p 2 --workload 1080p --pipeline --steps 100 --check -- wg7 product
bash profiles/r06_scripts/r06_ab_variants.sh r06s24_8k 2 --workload 8k --pipeline --steps 100 --check -- wg7 product
