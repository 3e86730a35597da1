set -x
TAG=${1:-r03b}
timeout 300 python -m pytest tests/test_pool.py tests/test_variants_gpu.py -m gpu -q 2>&1 | tail -3
bash tests/run_gpu_ab3.sh $TAG 2
for v in clocks clocks_persist; do
  MEAO_LIB_PATH=$PWD/miniengineao_amd/lib/variants/libmeao_$v.so timeout 300 python tools/phase_clocks.py > gpurun_out/phase_${v}_plain_$TAG.json 2>&1
  MEAO_LIB_PATH=$PWD/miniengineao_amd/lib/variants/libmeao_$v.so timeout 300 python tools/phase_clocks.py --pipeline > gpurun_out/phase_${v}_pipelined_$TAG.json 2>&1
done
python - <<PY
import json,glob
for f in sorted(glob.glob("gpurun_out/phase_*_$TAG.json")):
    try: d=json.load(open(f))
    except Exception as e: print(f, "unreadable", e); continue
    print(f.split("/")[-1], d.get("pass_us"))
    print("   ", {k.split()[0]+k.split()[1][:6]: v["us_per_wave"] for k,v in d.items() if isinstance(v,dict) and "us_per_wave" in v}, "sum", d.get("sum_us_per_wave (all upsample passes pooled)"))
PY
timeout 200 miniengineao_amd/lib/ubench_issue 4.0 bilateral > gpurun_out/ubench_bilateral_$TAG.txt 2>&1; cat gpurun_out/ubench_bilateral_$TAG.txt
