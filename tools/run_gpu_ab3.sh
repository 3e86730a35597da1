# round 3 A/B: per-pass times of the product and every variant library (not the clocks builds), plain and pipelined,
# alternating, ROUNDS rounds; then optional extras.   usage: bash tools/run_gpu_ab3.sh <tag> [rounds]
TAG=${1:-ab}; ROUNDS=${2:-2}
mkdir -p gpurun_out; OUT=gpurun_out/ab_$TAG.jsonl; : > $OUT
LIBS="product $(ls miniengineao_amd/lib/variants/libmeao_*.so 2>/dev/null | grep -v clocks)"
for r in $(seq $ROUNDS); do
  for lib in $LIBS; do
    if [ "$lib" = product ]; then unset MEAO_LIB_PATH; else export MEAO_LIB_PATH=$PWD/$lib; fi
    timeout 200 python tools/bench_passes.py --check ${BENCH_PASSES_ARGS} >> $OUT 2>>gpurun_out/ab_err_$TAG.log
    timeout 200 python tools/bench_passes.py --check --pipeline ${BENCH_PASSES_ARGS} >> $OUT 2>>gpurun_out/ab_err_$TAG.log
  done
done
unset MEAO_LIB_PATH
python - <<PY
import json
rows=[json.loads(l) for l in open("$OUT") if l.startswith("{")]
keys=["downsample","render","upsample_L3_to_L2","upsample_L2_to_L1","upsample_L1_to_L0","step_us","Gpix_s","ok"]
print("%-28s %-5s "%("lib","pipe")+" ".join("%9s"%k[-9:] for k in keys))
for r in rows:
    print("%-28s %-5s "%(r["tag"][:28], "pipe" if "downsample" not in r else "plain")+" ".join("%9s"%r.get(k,"") for k in keys))
PY
