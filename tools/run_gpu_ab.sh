# A/B on one box: bench.py with two flag sets, alternating, 3 runs each.   bash tools/run_gpu_ab.sh TAG "flagsA" "flagsB"
TAG=$1; A=$2; B=$3
mkdir -p gpurun_out
for i in 1 2 3; do
  timeout 300 python bench.py --no-cpu-baseline --skip-latency --no-copy-ceiling --no-best-host-config --validate-frames 1 $A 2>/dev/null | grep '^{' > gpurun_out/ab_${TAG}_A$i.json
  timeout 300 python bench.py --no-cpu-baseline --skip-latency --no-copy-ceiling --no-best-host-config --validate-frames 1 $B 2>/dev/null | grep '^{' > gpurun_out/ab_${TAG}_B$i.json
done
python - <<PY
import json,glob
for arm in "AB":
    for f in sorted(glob.glob("gpurun_out/ab_${TAG}_%s?.json" % arm)):
        try:
            j=json.loads(open(f).read())
            ps={p["kernel"]:p["ms"] for p in j["roofline"]["passes"]}
            print(arm, f.split("_")[-1], j["value"], j["ms_per_step"], j["validation"]["mismatching_frames"], {k:round(v*1e3,1) for k,v in ps.items()}, "plain", j["plain_launch_sequence"] and j["plain_launch_sequence"]["value"])
        except Exception as e:
            print(arm, f, "ERR", e)
PY
