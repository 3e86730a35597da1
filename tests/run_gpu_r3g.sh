TAG=${1:-r03g}
mkdir -p gpurun_out
for v in p2clocks p3clocks; do
MEAO_LIB_PATH=$PWD/miniengineao_amd/lib/variants/libmeao_$v.so timeout 300 python tools/phase_clocks.py 2>/dev/null > gpurun_out/phase_${v}_$TAG.json
python - <<PY
import json
d=json.load(open("gpurun_out/phase_${v}_$TAG.json"))
print("$v", d["pass_us"])
for lab in ("full_resolution_pass","persistent_loop"):
    print("  ",lab, {k[:18]: (v["us_per_wave"], v["waves"]) for k,v in d[lab].items() if isinstance(v,dict)}, d[lab]["sum_us_per_wave"])
PY
done
