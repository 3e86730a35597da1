# round 4, GPU call B: full GPU suite on the new kernels; integer-stripped blend kernels vs the r03 library; lean side-stream sweep
set -x
mkdir -p gpurun_out
T=r04b
timeout 1200 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_$T.log 2>&1; echo rc=$? >> gpurun_out/pytest_$T.log
tail -5 gpurun_out/pytest_$T.log
V=$PWD/miniengineao_amd/lib/variants
: > gpurun_out/ab_int_$T.jsonl
for r in 1 2 3; do
  for lib in product r03; do
    if [ $lib = product ]; then unset MEAO_LIB_PATH; else export MEAO_LIB_PATH=$V/libmeao_$lib.so; fi
    timeout 200 python tests/bench_passes.py --check --tag $lib-plain >> gpurun_out/ab_int_$T.jsonl 2>> gpurun_out/ab_err_$T.log
    timeout 200 python tests/bench_passes.py --check --pipeline --tag $lib-pipe >> gpurun_out/ab_int_$T.jsonl 2>> gpurun_out/ab_err_$T.log
  done
done
unset MEAO_LIB_PATH
cat gpurun_out/ab_int_$T.jsonl
: > gpurun_out/ab_side_$T.jsonl
for r in 1 2; do for m in 0 4 14 24 34 3 13 104 204; do
  if [ $m = 0 ]; then X=""; else X="--debug-set DS_SIDE_STREAM=$m"; fi
  timeout 200 python tests/bench_passes.py --pipeline --check $X >> gpurun_out/ab_side_$T.jsonl 2>> gpurun_out/ab_err_$T.log
done; done
cat gpurun_out/ab_side_$T.jsonl
# one frame per call: new three-level kernel vs r03
for lib in product r03 product r03; do
  if [ $lib = product ]; then unset MEAO_LIB_PATH; else export MEAO_LIB_PATH=$V/libmeao_$lib.so; fi
  for wl in 4k 1080p; do
  timeout 300 python bench.py --workload $wl --no-cpu-baseline --no-other-workloads --no-best-host-config --no-copy-ceiling --validate-frames 1 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$lib', '$wl', d['value'], d['ms_per_step'], d['single_frame'], d['validation']['mismatching_frames'])" >> gpurun_out/single_frame_$T.txt
  done
done
unset MEAO_LIB_PATH
cat gpurun_out/single_frame_$T.txt
tail -3 gpurun_out/ab_err_$T.log
