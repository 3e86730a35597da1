set -x
TAG=${1:-r03f}
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_variants_gpu.py -m gpu -q 2>&1 | tail -5
mkdir -p /tmp/hold; mv miniengineao_amd/lib/variants/libmeao_p2clocks.so /tmp/hold/
BENCH_PASSES_ARGS="" bash tests/run_gpu_ab3.sh $TAG 1
for lib in product $(ls miniengineao_amd/lib/variants/libmeao_*.so); do
  if [ "$lib" = product ]; then unset MEAO_LIB_PATH; else export MEAO_LIB_PATH=$PWD/$lib; fi
  timeout 200 python tests/bench_passes.py --check >> gpurun_out/ab2_$TAG.jsonl 2>/dev/null
done
unset MEAO_LIB_PATH
cat gpurun_out/ab2_$TAG.jsonl
MEAO_LIB_PATH=/tmp/hold/libmeao_p2clocks.so timeout 300 python tools/phase_clocks.py 2>/dev/null > gpurun_out/phase_persist2_plain_$TAG.json
python - <<PY
import json
d=json.load(open("gpurun_out/phase_persist2_plain_$TAG.json"))
print(d["pass_us"])
for lab in ("full_resolution_pass","blend_passes"):
    print("  ",lab, {k[:14]: v["us_per_wave"] for k,v in d[lab].items() if isinstance(v,dict)}, d[lab]["sum_us_per_wave"])
PY
