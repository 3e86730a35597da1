# round 4, GPU call P: LowDepth1 written with cacheable stores by the carried downsample tile (variant) vs non-temporal (product)
set -x
mkdir -p gpurun_out
T=r04p
V=$PWD/miniengineao_amd/lib/variants
: > gpurun_out/ab_low1_temporal_$T.jsonl
for r in 1 2 3; do
  timeout 200 python tests/bench_passes.py --pipeline --check --tag product >> gpurun_out/ab_low1_temporal_$T.jsonl 2>> gpurun_out/ab_err_$T.log
  MEAO_LIB_PATH=$V/libmeao_low1t.so timeout 200 python tests/bench_passes.py --pipeline --check --tag low1-cacheable >> gpurun_out/ab_low1_temporal_$T.jsonl 2>> gpurun_out/ab_err_$T.log
done
cat gpurun_out/ab_low1_temporal_$T.jsonl
