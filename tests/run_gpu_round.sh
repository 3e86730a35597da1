# Full GPU round: smoke, parity tests, bench line, rocprofv3 kernel stats, PMC passes.
set -x
TAG=${1:-r01}
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke_$TAG.log 2>&1; echo smoke rc=$? >> gpurun_out/smoke_$TAG.log
timeout 900 python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu_$TAG.log 2>&1; echo pytest rc=$? >> gpurun_out/pytest_gpu_$TAG.log
timeout 600 python bench.py 2>/dev/null | grep '^{' > gpurun_out/bench_$TAG.json
timeout 600 python bench.py --workload 1080p 2>/dev/null | grep '^{' > gpurun_out/bench_${TAG}_1080p.json
timeout 600 python bench.py --workload 8k 2>/dev/null | grep '^{' > gpurun_out/bench_${TAG}_8k.json
bash tests/run_rocprof.sh $TAG > gpurun_out/rocprof_$TAG.log 2>&1
bash tests/run_pmc.sh $TAG > gpurun_out/pmc_$TAG.log 2>&1
tail -3 gpurun_out/smoke_$TAG.log; tail -4 gpurun_out/pytest_gpu_$TAG.log; cut -c1-400 gpurun_out/bench_$TAG.json
