# round 4, GPU call T: the UNORM8 estimate's margin at 1.25 * 2^-11 (variant; proven bound 5.63e-4 of a code) against 2^-10 (product)
set -x
mkdir -p gpurun_out
T=r04t
V=$PWD/miniengineao_amd/lib/variants
OUT=gpurun_out/ab_r8_margin_$T.jsonl
: > $OUT
for r in 1 2 3 4; do
  timeout 200 python tests/bench_passes.py --pipeline --check --tag product >> $OUT 2>> gpurun_out/ab_err_$T.log
  MEAO_LIB_PATH=$V/libmeao_margin.so timeout 200 python tests/bench_passes.py --pipeline --check --tag margin >> $OUT 2>> gpurun_out/ab_err_$T.log
done
cat $OUT | cut -c1-300
MEAO_LIB_PATH=$V/libmeao_margin.so timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -2
