# round 4, GPU call C: side-stream split sweep, tall render tiles, pool enqueue cost
set -x
mkdir -p gpurun_out
T=r04c
timeout 900 python -m pytest tests/test_gpu_more.py -m gpu -q -x -k "side_stream or render_tilings" > gpurun_out/pytest_$T.log 2>&1; echo rc=$? >> gpurun_out/pytest_$T.log
tail -3 gpurun_out/pytest_$T.log
: > gpurun_out/ab_side_$T.jsonl
for r in 1 2; do for m in 0 4 3004 4004 5004 6004 7004 25004 5024; do
  if [ $m = 0 ]; then X=""; else X="--debug-set DS_SIDE_STREAM=$m"; fi
  timeout 200 python tests/bench_passes.py --pipeline --check $X >> gpurun_out/ab_side_$T.jsonl 2>> gpurun_out/ab_err_$T.log
done; done
cat gpurun_out/ab_side_$T.jsonl
: > gpurun_out/ab_tall_$T.jsonl
for r in 1 2 3; do for m in 0 64; do
  timeout 200 python tests/bench_passes.py --check --debug-set RENDER_TILE_H=$m >> gpurun_out/ab_tall_$T.jsonl 2>> gpurun_out/ab_err_$T.log
done; done
cat gpurun_out/ab_tall_$T.jsonl
timeout 300 python tools/pool_enqueue_cost.py > gpurun_out/pool_enqueue_cost_$T.jsonl 2>> gpurun_out/ab_err_$T.log
cat gpurun_out/pool_enqueue_cost_$T.jsonl
grep -v amdgpu.ids gpurun_out/ab_err_$T.log | tail -5
