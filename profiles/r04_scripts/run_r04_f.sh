# round 4, GPU call F2: render as persistent producer / consumer workgroups, 96x32 and 96x48 tiles
set -x
mkdir -p gpurun_out
T=r04f2
timeout 900 python -m pytest tests/test_gpu_more.py -m gpu -q -x -k "render_tilings" > gpurun_out/pytest_$T.log 2>&1; echo rc=$? >> gpurun_out/pytest_$T.log
tail -3 gpurun_out/pytest_$T.log
: > gpurun_out/ab_pc_$T.jsonl
for r in 1 2; do for m in 0 1 2; do
  timeout 200 python tests/bench_passes.py --check --debug-set RENDER_PRODUCER_CONSUMER=$m >> gpurun_out/ab_pc_$T.jsonl 2>> gpurun_out/ab_err_$T.log
done; done
cat gpurun_out/ab_pc_$T.jsonl
grep -v amdgpu.ids gpurun_out/ab_err_$T.log | tail -5
