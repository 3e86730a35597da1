# round 4, GPU call S: render term constants loaded in one batch before the window fill + a.dst[frame] fetched at the top of the upsample
# tile (product) against the previous library (prev), alternating
set -x
mkdir -p gpurun_out
T=r04s
V=$PWD/miniengineao_amd/lib/variants
OUT=gpurun_out/ab_scalar_loads_$T.jsonl
: > $OUT
for r in 1 2 3 4; do
  MEAO_LIB_PATH=$V/libmeao_prev.so timeout 200 python tests/bench_passes.py --pipeline --check --tag prev >> $OUT 2>> gpurun_out/ab_err_$T.log
  timeout 200 python tests/bench_passes.py --pipeline --check --tag product >> $OUT 2>> gpurun_out/ab_err_$T.log
done
cat $OUT | cut -c1-300
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -2
