# round 4, GPU call U: the hoisted AO quads of the blend passes held as one integer each and decoded behind the last barrier (product)
# against the previous library (prev: the quads were split into bytes, and waited for, in front of the first barrier)
set -x
mkdir -p gpurun_out
T=r04u
V=$PWD/miniengineao_amd/lib/variants
OUT=gpurun_out/ab_blend_ao_quads_$T.jsonl
: > $OUT
for r in 1 2 3 4; do
  MEAO_LIB_PATH=$V/libmeao_prev.so timeout 200 python tests/bench_passes.py --pipeline --check --tag prev >> $OUT 2>> gpurun_out/ab_err_$T.log
  timeout 200 python tests/bench_passes.py --pipeline --check --tag product >> $OUT 2>> gpurun_out/ab_err_$T.log
done
cat $OUT | cut -c1-300
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
