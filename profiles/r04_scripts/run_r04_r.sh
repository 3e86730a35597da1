# round 4, GPU call R: with the whole-tile bilateral phase in the product -- the forms of the bilateral texel in the kernel that carries
# the next downsample: exact sequences (product), UNORM8 estimate (fb1), grouped reciprocals (fb2), both (fb3); and the round-3 form
# without the whole-tile copy (nowt).  Then the GPU suite and a fuzz run on the product.
set -x
mkdir -p gpurun_out
T=r04r
V=$PWD/miniengineao_amd/lib/variants
OUT=gpurun_out/ab_fused_bilateral_forms_$T.jsonl
: > $OUT
for r in 1 2 3; do
  timeout 200 python tests/bench_passes.py --pipeline --check --tag product >> $OUT 2>> gpurun_out/ab_err_$T.log
  for v in fb1 fb2 fb3 nowt; do
    MEAO_LIB_PATH=$V/libmeao_$v.so timeout 200 python tests/bench_passes.py --pipeline --check --tag $v >> $OUT 2>> gpurun_out/ab_err_$T.log
  done
done
cat $OUT | cut -c1-400
timeout 200 python tests/bench_passes.py --check --tag product-plain | cut -c1-400
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu_$T.log 2>&1
tail -3 gpurun_out/pytest_gpu_$T.log
timeout 300 python tests/fuzz_gpu.py 200 93000 > gpurun_out/fuzz_$T.log 2>&1
tail -2 gpurun_out/fuzz_$T.log
