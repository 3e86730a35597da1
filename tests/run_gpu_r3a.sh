# round 3, call A: everything new on the CPU side + diagnostics for the upsample work
set -x
TAG=${1:-r03a}
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke_$TAG.log 2>&1; echo smoke rc=$? >> gpurun_out/smoke_$TAG.log
( time timeout 1500 python -m pytest tests -m gpu -q ) > gpurun_out/pytest_gpu_$TAG.log 2>&1; echo pytest rc=$? >> gpurun_out/pytest_gpu_$TAG.log
( time timeout 900 python bench.py --steps 20 --warmup 5 ) > gpurun_out/bench_full_$TAG.log 2>&1
grep '^{' gpurun_out/bench_full_$TAG.log > gpurun_out/bench_$TAG.json
timeout 600 python bench.py --pool 2 --no-cpu-baseline > gpurun_out/bench_pool2_$TAG.log 2>&1
timeout 600 python bench.py --pool 2 --no-pipeline > gpurun_out/bench_pool2_plain_$TAG.log 2>&1
V=$PWD/miniengineao_amd/lib/variants/libmeao_clocks.so
MEAO_LIB_PATH=$V timeout 300 python tools/phase_clocks.py > gpurun_out/phase_clocks_plain_$TAG.json 2>&1
MEAO_LIB_PATH=$V timeout 300 python tools/phase_clocks.py --pipeline > gpurun_out/phase_clocks_pipelined_$TAG.json 2>&1
timeout 200 miniengineao_amd/lib/ubench_issue 4.0 rcp > gpurun_out/ubench_rcp_$TAG.txt 2>&1
timeout 200 miniengineao_amd/lib/ubench_issue 4.0 bilateral >> gpurun_out/ubench_rcp_$TAG.txt 2>&1
timeout 200 miniengineao_amd/lib/ubench_issue 4.0 "v_fma_f32" >> gpurun_out/ubench_rcp_$TAG.txt 2>&1
tail -3 gpurun_out/smoke_$TAG.log; tail -6 gpurun_out/pytest_gpu_$TAG.log; cut -c1-600 gpurun_out/bench_$TAG.json; tail -2 gpurun_out/bench_pool2_$TAG.log | cut -c1-800
cat gpurun_out/phase_clocks_plain_$TAG.json | tail -30; cat gpurun_out/ubench_rcp_$TAG.txt
