# round 4, GPU call J: where the cost of the downsample rows carried by render goes (diagnostic variants: wrong results, timing only)
set -x
mkdir -p gpurun_out
T=r04j
V=$PWD/miniengineao_amd/lib/variants
: > gpurun_out/ab_dsr_diag_$T.jsonl
for r in 1 2; do
  timeout 200 python tests/bench_passes.py --pipeline --tag in-render --debug-set DS_IN_RENDER=1 >> gpurun_out/ab_dsr_diag_$T.jsonl 2>> gpurun_out/ab_err_$T.log
  MEAO_LIB_PATH=$V/libmeao_dsr1.so timeout 200 python tests/bench_passes.py --pipeline --tag loads-only --debug-set DS_IN_RENDER=1 >> gpurun_out/ab_dsr_diag_$T.jsonl 2>> gpurun_out/ab_err_$T.log
  MEAO_LIB_PATH=$V/libmeao_dsr2.so timeout 200 python tests/bench_passes.py --pipeline --tag arithmetic-and-stores-only --debug-set DS_IN_RENDER=1 >> gpurun_out/ab_dsr_diag_$T.jsonl 2>> gpurun_out/ab_err_$T.log
  timeout 200 python tests/bench_passes.py --tag plain >> gpurun_out/ab_dsr_diag_$T.jsonl 2>> gpurun_out/ab_err_$T.log
done
cat gpurun_out/ab_dsr_diag_$T.jsonl
grep -v amdgpu.ids gpurun_out/ab_err_$T.log | tail -5
