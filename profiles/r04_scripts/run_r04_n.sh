# round 4, GPU call N: the UNORM8 estimate (without reuse of its reciprocals on the exact path: no spills) in the fused last kernel
set -x
mkdir -p gpurun_out
T=r04n
V=$PWD/miniengineao_amd/lib/variants
MEAO_LIB_PATH=$V/libmeao_flest2.so timeout 600 python tests/variant_smoke.py > gpurun_out/variant_smoke_$T.log 2>&1; tail -1 gpurun_out/variant_smoke_$T.log
: > gpurun_out/ab_fused_est_$T.jsonl
for r in 1 2 3 4; do
  timeout 200 python tests/bench_passes.py --pipeline --check --tag product >> gpurun_out/ab_fused_est_$T.jsonl 2>> gpurun_out/ab_err_$T.log
  MEAO_LIB_PATH=$V/libmeao_flest2.so timeout 200 python tests/bench_passes.py --pipeline --check --tag fused-estimate-no-reuse >> gpurun_out/ab_fused_est_$T.jsonl 2>> gpurun_out/ab_err_$T.log
done
cat gpurun_out/ab_fused_est_$T.jsonl
