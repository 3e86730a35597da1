# round 4, GPU call E: experiment B (late depth window, 8 workgroups per CU) A/B, alone and with the side-stream downsample
set -x
mkdir -p gpurun_out
T=r04e
timeout 900 python -m pytest tests/test_gpu_more.py -m gpu -q -x -k "late_depth" > gpurun_out/pytest_$T.log 2>&1; echo rc=$? >> gpurun_out/pytest_$T.log
tail -3 gpurun_out/pytest_$T.log
: > gpurun_out/ab_late_$T.jsonl
for r in 1 2 3; do
  timeout 200 python tests/bench_passes.py --check --tag plain-7wg >> gpurun_out/ab_late_$T.jsonl 2>> gpurun_out/ab_err_$T.log
  timeout 200 python tests/bench_passes.py --check --tag plain-8wg --debug-set FINAL_LATE_DEPTH=1 >> gpurun_out/ab_late_$T.jsonl 2>> gpurun_out/ab_err_$T.log
  timeout 200 python tests/bench_passes.py --check --pipeline --tag side4-7wg --debug-set DS_SIDE_STREAM=4 >> gpurun_out/ab_late_$T.jsonl 2>> gpurun_out/ab_err_$T.log
  timeout 200 python tests/bench_passes.py --check --pipeline --tag side4-8wg --debug-set DS_SIDE_STREAM=4 --debug-set FINAL_LATE_DEPTH=1 >> gpurun_out/ab_late_$T.jsonl 2>> gpurun_out/ab_err_$T.log
done
cat gpurun_out/ab_late_$T.jsonl
cd /tmp && export TMPDIR=/tmp
for v in 0 1; do
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_${T}_late$v -o pmc -- python $GRAFT_REPO_ROOT/tests/bench_passes.py --steps 3 --debug-set FINAL_LATE_DEPTH=$v > $GRAFT_REPO_ROOT/gpurun_out/pmc_${T}_late$v.log 2>&1
done
cd $GRAFT_REPO_ROOT
python tests/pmc_summary.py gpurun_out/pmc_${T}_late0 2>/dev/null | grep "upsample_kernel<0, false, true" | head -3
python tests/pmc_summary.py gpurun_out/pmc_${T}_late1 2>/dev/null | grep "late_depth" | head -3
find gpurun_out -name '*kernel_trace.csv' -path "*pmc_${T}*" -delete
grep -v amdgpu.ids gpurun_out/ab_err_$T.log | tail -5
