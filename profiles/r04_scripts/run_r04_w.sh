# round 4, GPU call W: the scalars a full-resolution tile's prologue reads fetched in batches (variant bs: 5 dependent scalar-load
# round trips in front of the first global load instead of 10) against the product
set -x
mkdir -p gpurun_out
T=r04w
V=$PWD/miniengineao_amd/lib/variants
OUT=gpurun_out/ab_batched_scalar_loads_$T.jsonl
: > $OUT
for r in 1 2 3 4 5; do
  timeout 200 python tests/bench_passes.py --pipeline --check --tag product >> $OUT 2>> gpurun_out/ab_err_$T.log
  MEAO_LIB_PATH=$V/libmeao_bs.so timeout 200 python tests/bench_passes.py --pipeline --check --tag bs >> $OUT 2>> gpurun_out/ab_err_$T.log
done
cat $OUT | cut -c1-300
