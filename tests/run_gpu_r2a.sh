# Round-2 first GPU call: parity (incl. hostile depth), the VALU issue-cost microbenchmark, the bench line
# (also through the torch.distributed / RCCL path with one rank).
set -x
TAG=${1:-r02a}
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke_$TAG.log 2>&1; echo smoke rc=$? >> gpurun_out/smoke_$TAG.log
timeout 1200 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_gpu_$TAG.log 2>&1; echo pytest rc=$? >> gpurun_out/pytest_gpu_$TAG.log
timeout 300 miniengineao_amd/lib/ubench_issue 5.0 > gpurun_out/ubench_issue_$TAG.txt 2>&1
(cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU --output-format csv -d $OLDPWD/gpurun_out/ubench_pmc_$TAG -o pmc -- $OLDPWD/miniengineao_amd/lib/ubench_issue 1.0 > $OLDPWD/gpurun_out/ubench_pmc_$TAG.log 2>&1)
timeout 600 python bench.py 2>gpurun_out/bench_$TAG.err | grep '^{' > gpurun_out/bench_$TAG.json
MEAO_FORCE_DIST=1 timeout 600 python bench.py --no-cpu-baseline --skip-latency --min-time-ms 100 > gpurun_out/bench_force_dist_$TAG.log 2>&1
tail -3 gpurun_out/smoke_$TAG.log; tail -6 gpurun_out/pytest_gpu_$TAG.log; cut -c1-600 gpurun_out/bench_$TAG.json; tail -3 gpurun_out/bench_$TAG.err; head -40 gpurun_out/ubench_issue_$TAG.txt
