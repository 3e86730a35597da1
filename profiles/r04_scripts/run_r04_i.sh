# round 4, GPU call I: the next batch's downsample pass inside the render launch (lean rows in the texel loop)
set -x
mkdir -p gpurun_out
T=r04i
timeout 900 python -m pytest tests/test_gpu_more.py -m gpu -q -x -k "inside_the_render" > gpurun_out/pytest_$T.log 2>&1; echo rc=$? >> gpurun_out/pytest_$T.log
tail -5 gpurun_out/pytest_$T.log
: > gpurun_out/ab_dsr_$T.jsonl
for r in 1 2 3; do
  timeout 200 python tests/bench_passes.py --pipeline --check --tag fused-final >> gpurun_out/ab_dsr_$T.jsonl 2>> gpurun_out/ab_err_$T.log
  timeout 200 python tests/bench_passes.py --pipeline --check --tag in-render --debug-set DS_IN_RENDER=1 >> gpurun_out/ab_dsr_$T.jsonl 2>> gpurun_out/ab_err_$T.log
  timeout 200 python tests/bench_passes.py --pipeline --check --tag side4 --debug-set DS_SIDE_STREAM=4 >> gpurun_out/ab_dsr_$T.jsonl 2>> gpurun_out/ab_err_$T.log
done
cat gpurun_out/ab_dsr_$T.jsonl
grep -v amdgpu.ids gpurun_out/ab_err_$T.log | tail -5
