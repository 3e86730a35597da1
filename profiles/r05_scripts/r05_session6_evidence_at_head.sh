# Round-5 evidence at HEAD, one box: smoke, the GPU suite, fuzz (standard + long + through three variant libraries), every bench
# configuration, rocprofv3 kernel stats and the PMC passes of the SAME default command (so that events, trace and counters agree).
set -x
TAG=r05
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${TAG}_smoke.log 2>&1; echo smoke rc=$? >> gpurun_out/${TAG}_smoke.log
timeout 1500 python -m pytest tests -m gpu -q --maxfail=40 > gpurun_out/${TAG}_pytest_gpu.log 2>&1; echo pytest rc=$? >> gpurun_out/${TAG}_pytest_gpu.log
tail -3 gpurun_out/${TAG}_pytest_gpu.log
( time timeout 600 python bench.py --steps 20 --warmup 5 ) 2> gpurun_out/${TAG}_bench_driver_form_time.log | grep '^{' > gpurun_out/${TAG}_bench_driver_form.json
timeout 600 python bench.py 2> /dev/null | grep '^{' > gpurun_out/${TAG}_bench_4k.json
timeout 600 python bench.py --workload 1080p --no-other-workloads 2>/dev/null | grep '^{' > gpurun_out/${TAG}_bench_1080p.json
timeout 600 python bench.py --workload 8k --no-other-workloads 2>/dev/null | grep '^{' > gpurun_out/${TAG}_bench_8k.json
timeout 600 python bench.py --side-stream 4 --no-cpu-baseline --no-other-workloads --no-best-host-config --skip-latency 2>/dev/null | grep '^{' > gpurun_out/${TAG}_bench_side_stream4.json
timeout 600 python bench.py --gpus 2 --dist-backend gloo --batch 1 --no-cpu-baseline --skip-latency --no-copy-ceiling 2>/dev/null | grep '^{' > gpurun_out/${TAG}_bench_two_ranks_one_frame_each_gloo.json
timeout 600 python bench.py --gpus 1 --launcher --no-cpu-baseline --skip-latency --no-other-workloads --no-best-host-config 2>/dev/null | grep '^{' > gpurun_out/${TAG}_bench_one_rank_through_launcher_rccl.json
timeout 600 python bench.py --pool 2 2>/dev/null | grep '^{' > gpurun_out/${TAG}_bench_pool2.json
timeout 600 python bench.py --pool 8 --batch 1 2>/dev/null | grep '^{' > gpurun_out/${TAG}_bench_pool8_one_frame_each.json
timeout 600 python bench.py --shaded --no-cpu-baseline --skip-latency --no-other-workloads 2>/dev/null | grep '^{' > gpurun_out/${TAG}_bench_shaded.json
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r05_bench_*.json")):
    try:
        d = json.load(open(f)); print(f, d["value"], d["ms_per_step"], d["steps"], d["validation"]["mismatching_frames"], d.get("roofline", {}).get("frac"), d.get("without_pass_events"))
    except Exception as e:
        print(f, "ERR", e)
PY
bash tools/run_rocprof.sh $TAG > gpurun_out/rocprof_$TAG.log 2>&1
python tools/rocprof_timed_region.py gpurun_out/prof_$TAG/trace_kernel_trace.csv 30
PMC_GROUPS="sq1 sq2 sq5 fetch write" bash tools/run_pmc.sh $TAG > gpurun_out/pmc_$TAG.log 2>&1
PMC_GROUPS="sq1 fetch write" bash tools/run_pmc.sh ${TAG}_1080p --workload 1080p > gpurun_out/pmc_${TAG}_1080p.log 2>&1
PMC_GROUPS="sq1 fetch write" bash tools/run_pmc.sh ${TAG}_8k --workload 8k > gpurun_out/pmc_${TAG}_8k.log 2>&1
timeout 600 python tools/fuzz_gpu.py 200 12000 > gpurun_out/${TAG}_fuzz_gpu.log 2>&1; tail -1 gpurun_out/${TAG}_fuzz_gpu.log
# (the 6000-case fuzz and the variant fuzz runs were made by the previous run of this script, same kernels: profiles/r05_fuzz_gpu_long.log, r05_fuzz_gpu_variant_*.log)
