# Round-5 GPU session 2: the suite at the final defaults, PMC passes at the current kernels (3 workloads), kernel stats of the
# timed region, what two pool members overlap (kernel trace), why the raw-depth render is not faster (FETCH_SIZE per launch), fuzz.
set -x
TAG=r05
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q --maxfail=40 > gpurun_out/${TAG}_pytest_gpu.log 2>&1; echo pytest rc=$? >> gpurun_out/${TAG}_pytest_gpu.log
tail -4 gpurun_out/${TAG}_pytest_gpu.log
timeout 900 python tools/fuzz_gpu.py 200 12000 > gpurun_out/${TAG}_fuzz_gpu.log 2>&1; tail -2 gpurun_out/${TAG}_fuzz_gpu.log
bash tools/run_rocprof.sh $TAG > gpurun_out/rocprof_$TAG.log 2>&1
PMC_GROUPS="sq1 sq2 sq5 fetch write" bash tools/run_pmc.sh $TAG > gpurun_out/pmc_$TAG.log 2>&1
PMC_GROUPS="sq1 fetch write" bash tools/run_pmc.sh ${TAG}_1080p --workload 1080p > gpurun_out/pmc_${TAG}_1080p.log 2>&1
PMC_GROUPS="sq1 fetch write" bash tools/run_pmc.sh ${TAG}_8k --workload 8k > gpurun_out/pmc_${TAG}_8k.log 2>&1
# plain launch sequence too (stand-alone downsample_kernel and plain upsample_kernel<final>: the calibration rows)
PMC_GROUPS="fetch write" bash tools/run_pmc.sh ${TAG}_plain --no-pipeline > gpurun_out/pmc_${TAG}_plain.log 2>&1
REPO=$PWD
# two pool members on one device: what overlaps
mkdir -p gpurun_out/prof_${TAG}_pool2 gpurun_out/prof_${TAG}_single gpurun_out/pmc_${TAG}_from_depth
( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $REPO/gpurun_out/prof_${TAG}_pool2 -o trace -- python $REPO/bench.py --pool 2 --steps 20 --warmup 5 > $REPO/gpurun_out/prof_${TAG}_pool2/bench.log 2>&1 )
python tools/kernel_overlap.py $(find gpurun_out/prof_${TAG}_pool2 -name '*kernel_trace.csv' | head -1) 320 > gpurun_out/${TAG}_kernel_overlap_pool2.txt 2>&1
python tools/kernel_overlap.py $(find gpurun_out/prof_${TAG} -name '*kernel_trace.csv' | head -1) 160 > gpurun_out/${TAG}_kernel_overlap_single_context.txt 2>&1
cat gpurun_out/${TAG}_kernel_overlap_pool2.txt gpurun_out/${TAG}_kernel_overlap_single_context.txt
# one 4K frame per call: bytes fetched per launch, stored mips vs raw depth
( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $REPO/gpurun_out/pmc_${TAG}_from_depth/fetch -o pmc -- python $REPO/tools/from_depth_sweep.py --workloads 4k --frames 1 --iters 20 --rounds 1 > $REPO/gpurun_out/pmc_${TAG}_from_depth/fetch.log 2>&1 )
python - <<'PY' > gpurun_out/r05_from_depth_fetch_bytes.txt 2>&1
import csv, glob, collections, re
by = collections.defaultdict(list)
for f in glob.glob("gpurun_out/pmc_r05_from_depth/fetch/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == "FETCH_SIZE":
            m = re.search(r"(\w+_kernel<[^>]*>)", r["Kernel_Name"])
            by[m.group(1) if m else r["Kernel_Name"][:40]].append(float(r["Counter_Value"]))
print("one 4K frame per call: bytes read per launch = FETCH_SIZE (KiB) x 1024 x 2 (profiles/r05_pmc_calibration.json)")
for k, v in sorted(by.items()):
    print(f"  {sum(v) / len(v) * 2048 / 1e6:9.2f} MB  x{len(v):4d}  {k}")
PY
cat gpurun_out/r05_from_depth_fetch_bytes.txt
timeout 600 python bench.py 2> /dev/null | grep '^{' > gpurun_out/${TAG}_bench_4k.json
timeout 600 python bench.py --steps 20 --warmup 5 2> /dev/null | grep '^{' > gpurun_out/${TAG}_bench_driver_form.json
find gpurun_out/prof_${TAG}_pool2 -name '*kernel_trace.csv' -size +20M -delete
python tools/rocprof_timed_region.py gpurun_out/prof_$TAG/trace_kernel_trace.csv 30
python - <<'PY'
import json
for f in ("gpurun_out/r05_bench_4k.json", "gpurun_out/r05_bench_driver_form.json"):
    d = json.load(open(f)); print(f, d["value"], d["ms_per_step"], d["steps"], d["validation"]["mismatching_frames"], d["roofline"]["frac"], d["single_frame"])
PY
