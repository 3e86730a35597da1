# round 4, GPU call O: all three blend passes in one launch for full batches (NESTED_MAX_TILES forced) vs the size-chosen two-level + L2->L1
set -x
mkdir -p gpurun_out
T=r04o
: > gpurun_out/ab_three_level_batches_$T.jsonl
for r in 1 2 3; do
  timeout 200 python tests/bench_passes.py --pipeline --check --tag two-level+L2toL1 >> gpurun_out/ab_three_level_batches_$T.jsonl 2>> gpurun_out/ab_err_$T.log
  timeout 200 python tests/bench_passes.py --pipeline --check --tag three-level --debug-set NESTED_MAX_TILES=1000000 >> gpurun_out/ab_three_level_batches_$T.jsonl 2>> gpurun_out/ab_err_$T.log
done
cat gpurun_out/ab_three_level_batches_$T.jsonl
