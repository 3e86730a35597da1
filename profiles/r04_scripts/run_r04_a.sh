# round 4, GPU call A: new bench paths + side-stream downsample A/B + PMC passes for 8K and 1080p
set -x
mkdir -p gpurun_out
T=r04a
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke_$T.log 2>&1; echo smoke rc=$? >> gpurun_out/smoke_$T.log
timeout 900 python -m pytest tests/test_bench_multirank_gpu.py tests/test_gpu_more.py -m gpu -q -x -k "side_stream or bench or ranks or plain_invocation or one_frame or rccl" > gpurun_out/pytest_$T.log 2>&1; echo rc=$? >> gpurun_out/pytest_$T.log
tail -5 gpurun_out/pytest_$T.log
# side-stream sweep, alternating, 2 rounds: 0 = fused last kernel (product)
: > gpurun_out/ab_side_$T.jsonl
for r in 1 2; do for m in 0 1 11 21 31 2 12 4; do
  if [ $m = 0 ]; then X=""; else X="--debug-set DS_SIDE_STREAM=$m"; fi
  timeout 200 python tests/bench_passes.py --pipeline --check $X >> gpurun_out/ab_side_$T.jsonl 2>> gpurun_out/ab_side_err_$T.log
done; done
cat gpurun_out/ab_side_$T.jsonl
( time timeout 600 python bench.py --steps 20 --warmup 5 ) 2> gpurun_out/bench_${T}_time.log | grep '^{' > gpurun_out/bench_$T.json
cut -c1-600 gpurun_out/bench_$T.json
PMC_GROUPS="sq1 fetch write" bash tests/run_pmc.sh ${T}_8k --workload 8k > gpurun_out/pmc_${T}_8k.log 2>&1
PMC_GROUPS="sq1 fetch write" bash tests/run_pmc.sh ${T}_1080p --workload 1080p > gpurun_out/pmc_${T}_1080p.log 2>&1
tail -3 gpurun_out/pmc_${T}_8k.log gpurun_out/pmc_${T}_1080p.log
