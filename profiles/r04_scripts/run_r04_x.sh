# round 4, GPU call X: MEAO_DEBUG_SPLIT_BATCH -- render and the blend launches of the second half of the batch on a second stream
# (1 = lowest priority, 2 = default priority) against the product; parity test of the option first
set -x
mkdir -p gpurun_out
T=r04x
OUT=gpurun_out/ab_split_batch_$T.jsonl
: > $OUT
timeout 300 python -m pytest tests/test_gpu_more.py -m gpu -x -q -k "halves_on_two_streams" 2>&1 | tail -3
for r in 1 2 3; do
  timeout 200 python tests/bench_passes.py --pipeline --check --tag product >> $OUT 2>> gpurun_out/ab_err_$T.log
  timeout 200 python tests/bench_passes.py --pipeline --check --debug-set SPLIT_BATCH=1 --tag split-low >> $OUT 2>> gpurun_out/ab_err_$T.log
  timeout 200 python tests/bench_passes.py --pipeline --check --debug-set SPLIT_BATCH=2 --tag split-equal >> $OUT 2>> gpurun_out/ab_err_$T.log
done
cat $OUT | cut -c1-300
tail -3 gpurun_out/ab_err_$T.log
