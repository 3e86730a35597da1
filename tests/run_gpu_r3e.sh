set -x
TAG=${1:-r03e}
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_variants_gpu.py -m gpu -q 2>&1 | tail -5
BENCH_PASSES_ARGS="" bash tests/run_gpu_ab3.sh $TAG 2
MEAO_LIB_PATH=$PWD/miniengineao_amd/lib/variants/libmeao_persist3c.so timeout 300 python tools/phase_clocks.py 2>/dev/null > gpurun_out/phase_persist3_plain_$TAG.json
python - <<PY
import json
d=json.load(open("gpurun_out/phase_persist3_plain_$TAG.json"))
print(d["pass_us"])
for lab in ("full_resolution_pass","blend_passes"):
    print("  ",lab, {k[:14]: v["us_per_wave"] for k,v in d[lab].items() if isinstance(v,dict)}, d[lab]["sum_us_per_wave"])
PY
