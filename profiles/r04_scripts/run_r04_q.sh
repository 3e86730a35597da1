# round 4, GPU call Q: the bilateral phase's arms -- uniform operands in VGPRs (bil_vc), a copy of the phase for tiles wholly inside
# the frame (bil_wt), both (bil_vcwt), both + v_cvt_pk_u8_f32 packing (bil_all) -- against the product, alternating, three rounds
set -x
mkdir -p gpurun_out
T=r04q
V=$PWD/miniengineao_amd/lib/variants
OUT=gpurun_out/ab_bilateral_arms_$T.jsonl
: > $OUT
MEAO_LIB_PATH=$V/libmeao_bil_all.so timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "exhaustive" > gpurun_out/selftest_bil_all_$T.log 2>&1
tail -3 gpurun_out/selftest_bil_all_$T.log
for r in 1 2 3; do
  timeout 200 python tests/bench_passes.py --pipeline --check --tag product >> $OUT 2>> gpurun_out/ab_err_$T.log
  for v in bil_vc bil_wt bil_vcwt bil_all; do
    MEAO_LIB_PATH=$V/libmeao_$v.so timeout 200 python tests/bench_passes.py --pipeline --check --tag $v >> $OUT 2>> gpurun_out/ab_err_$T.log
  done
done
cat $OUT | cut -c1-400
MEAO_LIB_PATH=$V/libmeao_bil_all.so timeout 600 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_bil_all_$T.log 2>&1
tail -3 gpurun_out/pytest_bil_all_$T.log
MEAO_LIB_PATH=$V/libmeao_bil_all.so timeout 300 python tests/fuzz_gpu.py 150 91000 > gpurun_out/fuzz_bil_all_$T.log 2>&1
tail -2 gpurun_out/fuzz_bil_all_$T.log
