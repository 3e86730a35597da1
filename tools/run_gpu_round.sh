# Full GPU round: smoke, parity tests, bench lines, rocprofv3 kernel stats, PMC passes, microbenchmarks.
set -x
TAG=${1:-r04}
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke_$TAG.log 2>&1; echo smoke rc=$? >> gpurun_out/smoke_$TAG.log
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu_$TAG.log 2>&1; echo pytest rc=$? >> gpurun_out/pytest_gpu_$TAG.log
( time timeout 600 python bench.py --steps 20 --warmup 5 ) 2> gpurun_out/bench_${TAG}_time.log | grep '^{' > gpurun_out/bench_$TAG.json
timeout 600 python bench.py --workload 1080p --no-other-workloads 2>/dev/null | grep '^{' > gpurun_out/bench_${TAG}_1080p.json
timeout 600 python bench.py --workload 8k --no-other-workloads 2>/dev/null | grep '^{' > gpurun_out/bench_${TAG}_8k.json
timeout 600 python bench.py --gpus 2 --dist-backend gloo --batch 1 --no-cpu-baseline --skip-latency --no-copy-ceiling > gpurun_out/bench_${TAG}_two_ranks_one_frame_each_gloo.log 2>&1
timeout 300 python tools/pool_enqueue_cost.py > gpurun_out/pool_enqueue_cost_$TAG.jsonl 2>/dev/null
timeout 600 python bench.py --pool 2 > gpurun_out/bench_${TAG}_pool2.log 2>&1
timeout 600 python bench.py --pool 3 --batch 8 > gpurun_out/bench_${TAG}_pool3.log 2>&1
MEAO_FORCE_DIST=1 timeout 600 python bench.py --no-cpu-baseline --skip-latency --no-other-workloads --min-time-ms 100 > gpurun_out/bench_force_dist_$TAG.log 2>&1
timeout 600 python bench.py --shaded --no-cpu-baseline --skip-latency --no-other-workloads 2>/dev/null | grep '^{' > gpurun_out/bench_${TAG}_shaded.json
timeout 600 python tools/fuzz_gpu.py 200 12000 > gpurun_out/fuzz_$TAG.log 2>&1
bash tools/run_rocprof.sh $TAG > gpurun_out/rocprof_$TAG.log 2>&1
bash tools/run_pmc.sh $TAG > gpurun_out/pmc_$TAG.log 2>&1
MEAO_LIB_PATH=$PWD/miniengineao_amd/lib/variants/libmeao_clocks.so timeout 300 python tools/phase_clocks.py 2>/dev/null > gpurun_out/phase_clocks_plain_$TAG.json
MEAO_LIB_PATH=$PWD/miniengineao_amd/lib/variants/libmeao_clocks.so timeout 300 python tools/phase_clocks.py --pipeline 2>/dev/null > gpurun_out/phase_clocks_pipelined_$TAG.json
timeout 300 miniengineao_amd/lib/ubench_issue 5.0 > gpurun_out/ubench_issue_$TAG.txt 2>&1
timeout 300 miniengineao_amd/lib/ubench_lds 4.0 > gpurun_out/ubench_lds_$TAG.txt 2>&1
timeout 120 miniengineao_amd/lib/ubench_launch > gpurun_out/ubench_launch_$TAG.txt 2>&1
tail -3 gpurun_out/smoke_$TAG.log; tail -4 gpurun_out/pytest_gpu_$TAG.log; cut -c1-400 gpurun_out/bench_$TAG.json
