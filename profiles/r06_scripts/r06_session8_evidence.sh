# Round 6, session 8: evidence lines of the LinearDepth-free library with its own counter file (profiles/pmc_traffic.json, hash-matched).
set -x
T=r06
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke_$T.log 2>&1; echo smoke rc=$? >> gpurun_out/smoke_$T.log
( time timeout 900 python bench.py --steps 20 --warmup 5 ) 2> gpurun_out/bench_${T}_time.log | grep '^{' > gpurun_out/bench_${T}_driver_form.json
timeout 900 python bench.py 2>/dev/null | grep '^{' > gpurun_out/bench_$T.json
timeout 600 python bench.py --workload 1080p --no-other-workloads 2>/dev/null | grep '^{' > gpurun_out/bench_${T}_1080p.json
timeout 600 python bench.py --workload 8k --no-other-workloads 2>/dev/null | grep '^{' > gpurun_out/bench_${T}_8k.json
timeout 600 python bench.py --shaded --no-cpu-baseline --skip-latency --no-other-workloads 2>/dev/null | grep '^{' > gpurun_out/bench_${T}_shaded.json
timeout 600 python bench.py --gpus 2 --dist-backend gloo --batch 1 --no-cpu-baseline --skip-latency --no-copy-ceiling 2>&1 | grep '^{' > gpurun_out/bench_${T}_two_ranks_one_frame_each_gloo.json
timeout 600 python bench.py --gpus 1 --launcher --no-cpu-baseline --skip-latency --no-other-workloads --no-best-host-config 2>&1 | grep '^{' > gpurun_out/bench_${T}_one_rank_through_launcher_rccl.json
timeout 600 python bench.py --pool 8 --batch 1 2>&1 | grep '^{' > gpurun_out/bench_${T}_pool8_one_frame_each.json
timeout 600 python bench.py --pool 2 2>&1 | grep '^{' > gpurun_out/bench_${T}_pool2.json
python bench.py --gpus 8 --dry-run-topology > gpurun_out/bench_${T}_dry_run_topology.json 2>/dev/null
timeout 1500 python -m pytest tests -m gpu -q -rs > gpurun_out/pytest_gpu_$T.log 2>&1; echo pytest rc=$? >> gpurun_out/pytest_gpu_$T.log
timeout 1500 python tools/fuzz_gpu.py 3000 26000 > gpurun_out/fuzz_$T.log 2>&1
tail -2 gpurun_out/smoke_$T.log; tail -4 gpurun_out/pytest_gpu_$T.log; tail -2 gpurun_out/fuzz_$T.log; cut -c1-300 gpurun_out/bench_${T}_driver_form.json
