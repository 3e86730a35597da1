# Round-5 evidence after the profiling events lost their system-scope fence (the kernels are the ones of r05_session6): the bench
# configurations whose figures the documents quote, and rocprofv3 kernel stats of the same default command on the same box.
set -x
TAG=r05ff
mkdir -p gpurun_out
( time timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 ) 2> gpurun_out/${TAG}_bench_driver_form_time.log | grep '^{' > gpurun_out/${TAG}_bench_driver_form.json
timeout 600 python bench.py 2> /dev/null | grep '^{' > gpurun_out/${TAG}_bench_4k.json
timeout 600 python bench.py --workload 1080p --no-other-workloads 2>/dev/null | grep '^{' > gpurun_out/${TAG}_bench_1080p.json
timeout 600 python bench.py --workload 8k --no-other-workloads 2>/dev/null | grep '^{' > gpurun_out/${TAG}_bench_8k.json
timeout 600 python bench.py --pool 2 2>/dev/null | grep '^{' > gpurun_out/${TAG}_bench_pool2.json
timeout 600 python bench.py --shaded --no-cpu-baseline --skip-latency --no-other-workloads 2>/dev/null | grep '^{' > gpurun_out/${TAG}_bench_shaded.json
timeout 600 python bench.py --gpus 1 --launcher --no-cpu-baseline --skip-latency --no-other-workloads --no-best-host-config 2>/dev/null | grep '^{' > gpurun_out/${TAG}_bench_one_rank_through_launcher_rccl.json
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r05ff_bench_*.json")):
    try:
        d = json.load(open(f)); r = d.get("roofline", {})
        print(f, d["value"], d["ms_per_step"], d["steps"], d["validation"]["mismatching_frames"], r.get("frac"), r.get("real_traffic_frac"), r.get("real_traffic_vs_copy_ceiling_frac"), [p["ms"] for p in r.get("passes", [])], (r.get("render_plus_upsample") or {}).get("frac"), d.get("without_pass_events"), (d.get("best_host_config") or {}).get("value"), d.get("single_frame_latency_ms"))
    except Exception as e:
        print(f, "ERR", e)
PY
bash tools/run_rocprof.sh $TAG > gpurun_out/rocprof_$TAG.log 2>&1
python tools/rocprof_timed_region.py gpurun_out/prof_$TAG/trace_kernel_trace.csv 30 | tee gpurun_out/${TAG}_kernel_trace_timed_region.txt
