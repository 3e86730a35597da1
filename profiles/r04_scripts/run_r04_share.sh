# round 4: part of the carried downsample in the L2->L1 blend launch (MEAO_DEBUG_DS_SHARE_IN_BLEND percent), re-measured now that the last
# kernel moves its bytes at 0.8 of the copy rate
set -x
mkdir -p gpurun_out
OUT=gpurun_out/ab_ds_share_in_blend_r04.jsonl
: > $OUT
for r in 1 2; do
  for sh in 0 10 20 35; do
    timeout 200 python tests/bench_passes.py --pipeline --check --ds-share $sh --tag share$sh >> $OUT 2>> gpurun_out/ab_err_share.log
  done
done
cat $OUT | cut -c1-300
