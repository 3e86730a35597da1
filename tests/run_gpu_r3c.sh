set -x
TAG=${1:-r03c}
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_more.py tests/test_pool.py -m gpu -q -k "carried_downsample_split or pool" 2>&1 | tail -3
for share in 0 15 30 45 0 15 30 45; do
  timeout 200 python tests/bench_passes.py --check --pipeline --ds-share $share >> gpurun_out/ab_share_$TAG.jsonl 2>>gpurun_out/ab_err_$TAG.log
done
cat gpurun_out/ab_share_$TAG.jsonl
MEAO_LIB_PATH=$PWD/miniengineao_amd/lib/variants/libmeao_clocks.so timeout 300 python tools/phase_clocks.py 2>/dev/null > gpurun_out/phase_clocks_plain_$TAG.json
MEAO_LIB_PATH=$PWD/miniengineao_amd/lib/variants/libmeao_clocks.so timeout 300 python tools/phase_clocks.py --pipeline 2>/dev/null > gpurun_out/phase_clocks_pipelined_$TAG.json
cat gpurun_out/phase_clocks_plain_$TAG.json gpurun_out/phase_clocks_pipelined_$TAG.json | grep -v waves
timeout 200 miniengineao_amd/lib/ubench_issue 4.0 bilateral > gpurun_out/ubench_bilateral_$TAG.txt 2>&1
timeout 200 miniengineao_amd/lib/ubench_issue 4.0 mix >> gpurun_out/ubench_bilateral_$TAG.txt 2>&1
cat gpurun_out/ubench_bilateral_$TAG.txt
