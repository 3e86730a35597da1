# per-pass times of every variant library under miniengineao_amd/lib/variants (two rounds, alternating)
mkdir -p gpurun_out; OUT=gpurun_out/variants_${1:-x}.jsonl; : > $OUT
for round in 1 2; do
  for lib in miniengineao_amd/lib/variants/libmeao_*.so; do
    MEAO_LIB_PATH=$PWD/$lib timeout 200 python tools/bench_passes.py --check ${BENCH_PASSES_ARGS} >> $OUT 2>>gpurun_out/variants_err.log
  done
done
cat $OUT
