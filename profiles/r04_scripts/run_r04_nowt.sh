# round 4: 1500 random configurations through the variant library without the whole-tile copy of the bilateral phase
# (every tile takes the masked form that partial tiles take in the product), and 1500 more through the product
set -x
mkdir -p gpurun_out
V=$PWD/miniengineao_amd/lib/variants
MEAO_LIB_PATH=$V/libmeao_nowt.so timeout 600 python tests/fuzz_gpu.py 1500 200000 > gpurun_out/fuzz_nowt_r04.log 2>&1
tail -2 gpurun_out/fuzz_nowt_r04.log
timeout 600 python tests/fuzz_gpu.py 1500 300000 > gpurun_out/fuzz_more_r04.log 2>&1
tail -2 gpurun_out/fuzz_more_r04.log
