# round 4, GPU call Y: the whole library compiled with another machine scheduler (-mllvm -misched=gcn-max-ilp /
# gcn-iterative-max-occupancy-experimental) against the product (default scheduler)
set -x
mkdir -p gpurun_out
T=r04y
V=$PWD/miniengineao_amd/lib/variants
OUT=gpurun_out/ab_machine_scheduler_$T.jsonl
: > $OUT
for r in 1 2 3; do
  timeout 200 python tests/bench_passes.py --pipeline --check --tag product >> $OUT 2>> gpurun_out/ab_err_$T.log
  for v in sch_maxilp sch_maxocc; do
    MEAO_LIB_PATH=$V/libmeao_$v.so timeout 200 python tests/bench_passes.py --pipeline --check --tag $v >> $OUT 2>> gpurun_out/ab_err_$T.log
  done
done
cat $OUT | cut -c1-300
