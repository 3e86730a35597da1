# round 4, GPU call M2: lean carried tile in the fused last kernel, with / without the UNORM8 estimate and grouped reciprocals
set -x
mkdir -p gpurun_out
T=r04m2
V=$PWD/miniengineao_amd/lib/variants
for v in fl_est fl_est_grp fl_grp; do MEAO_LIB_PATH=$V/libmeao_$v.so timeout 600 python tests/variant_smoke.py > gpurun_out/variant_smoke_${v}_$T.log 2>&1; tail -1 gpurun_out/variant_smoke_${v}_$T.log; done
: > gpurun_out/ab_fusedlean_$T.jsonl
for r in 1 2 3; do
  timeout 200 python tests/bench_passes.py --pipeline --check --tag product >> gpurun_out/ab_fusedlean_$T.jsonl 2>> gpurun_out/ab_err_$T.log
  for v in fusedlean fl_est fl_est_grp fl_grp; do
    MEAO_LIB_PATH=$V/libmeao_$v.so timeout 200 python tests/bench_passes.py --pipeline --check --tag $v >> gpurun_out/ab_fusedlean_$T.jsonl 2>> gpurun_out/ab_err_$T.log
  done
done
cat gpurun_out/ab_fusedlean_$T.jsonl
