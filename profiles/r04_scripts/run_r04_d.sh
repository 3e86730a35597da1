# round 4, GPU call D: full suite with the pool's worker threads, pool enqueue cost, bench line with best_host_config, shape-4 side kernel, rocprof + PMC of the r04 kernels
set -x
mkdir -p gpurun_out
T=r04d
timeout 1200 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_$T.log 2>&1; echo rc=$? >> gpurun_out/pytest_$T.log
tail -4 gpurun_out/pytest_$T.log
timeout 300 python tools/pool_enqueue_cost.py > gpurun_out/pool_enqueue_cost_$T.jsonl 2>> gpurun_out/ab_err_$T.log
cat gpurun_out/pool_enqueue_cost_$T.jsonl
: > gpurun_out/ab_side_$T.jsonl
for r in 1 2; do for m in 0 4 44; do
  if [ $m = 0 ]; then X=""; else X="--debug-set DS_SIDE_STREAM=$m"; fi
  timeout 200 python tests/bench_passes.py --pipeline --check $X >> gpurun_out/ab_side_$T.jsonl 2>> gpurun_out/ab_err_$T.log
done; done
cat gpurun_out/ab_side_$T.jsonl
( time timeout 600 python bench.py --steps 20 --warmup 5 ) 2> gpurun_out/bench_${T}_time.log | grep '^{' > gpurun_out/bench_$T.json
python -c "
import json; d=json.load(open('gpurun_out/bench_$T.json')); print(d['value'], d['ms_per_step'], d['best_host_config'])"
timeout 300 python bench.py --side-stream 4 --no-cpu-baseline --no-other-workloads --no-best-host-config --skip-latency 2>/dev/null | grep '^{' > gpurun_out/bench_${T}_side4.json
cut -c1-300 gpurun_out/bench_${T}_side4.json
bash tests/run_rocprof.sh $T > gpurun_out/rocprof_$T.log 2>&1
PMC_GROUPS="sq1 sq2 sq5 fetch write" bash tests/run_pmc.sh $T > gpurun_out/pmc_$T.log 2>&1
tail -30 gpurun_out/pmc_$T.log | grep -v "^+"
grep -v amdgpu.ids gpurun_out/ab_err_$T.log | tail -5
