# Round-5 GPU session 7: the R8 result stream reports 1.31x its payload in WRITE_SIZE -- streaming (nt) half-line stores?  Product vs the
# `plainstore` variant (temporal stores): WRITE_SIZE per launch and an alternating A/B.
set -x
mkdir -p gpurun_out
REPO=$PWD
for lib in "" miniengineao_amd/lib/variants/libmeao_plainstore.so; do
  name=${lib##*/}; name=${name:-product}
  OUT=$REPO/gpurun_out/pmc_r05_stores_$name; mkdir -p $OUT
  ( cd /tmp && export TMPDIR=/tmp && MEAO_LIB_PATH=${lib:+$REPO/$lib} timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -o pmc -- python $REPO/bench.py --no-cpu-baseline --skip-latency --no-other-workloads --no-copy-ceiling --no-best-host-config --validate-frames 0 --min-time-ms 0 --steps 3 --warmup 1 > $OUT/write.log 2>&1 )
  python tools/pmc_summary.py $OUT 2>/dev/null | grep -E "upsample_kernel<0, false, true|upsample_final_with"
done
rm -f gpurun_out/r05_ab_result_stores.jsonl
for i in 1 2 3; do for lib in "" miniengineao_amd/lib/variants/libmeao_plainstore.so; do
MEAO_LIB_PATH=${lib:+$PWD/$lib} timeout 300 python bench.py --no-cpu-baseline --skip-latency --no-other-workloads --no-best-host-config --no-copy-ceiling --validate-frames 2 2>/dev/null | grep "^{" | python -c "
import sys,json
d=json.loads(sys.stdin.read()); ps={p['kernel']:round(p['ms']*1e3,1) for p in d['roofline']['passes']}
print(json.dumps({'lib':'${lib##*/}' or 'product','value':d['value'],'ms_per_step':d['ms_per_step'],'passes':ps,'plain_pass_ms':d['plain_launch_sequence']['pass_ms'],'mismatching':d['validation']['mismatching_frames']}))" >> gpurun_out/r05_ab_result_stores.jsonl
done; done
cat gpurun_out/r05_ab_result_stores.jsonl | cut -c1-420
