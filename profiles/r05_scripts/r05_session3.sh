# Round-5 GPU session 3: bench lines at the final defaults (the session-2 lines carried a headline bug of bench.py: a loop
# variable named `value`), write-side counter calibration on image-shaped streams, cache counters of the raw-depth render, a long fuzz
# that includes the raw-depth forms, the other bench configurations.
set -x
TAG=r05
mkdir -p gpurun_out
REPO=$PWD
( time timeout 600 python bench.py --steps 20 --warmup 5 ) 2> gpurun_out/${TAG}_bench_driver_form_time.log | grep '^{' > gpurun_out/${TAG}_bench_driver_form.json
timeout 600 python bench.py 2> /dev/null | grep '^{' > gpurun_out/${TAG}_bench_4k.json
timeout 600 python bench.py --workload 1080p --no-other-workloads 2>/dev/null | grep '^{' > gpurun_out/${TAG}_bench_1080p.json
timeout 600 python bench.py --workload 8k --no-other-workloads 2>/dev/null | grep '^{' > gpurun_out/${TAG}_bench_8k.json
timeout 600 python bench.py --side-stream 4 --no-cpu-baseline --no-other-workloads --no-best-host-config --skip-latency 2>/dev/null | grep '^{' > gpurun_out/${TAG}_bench_side_stream4.json
timeout 600 python bench.py --batch 32 --no-cpu-baseline --no-other-workloads --no-best-host-config --skip-latency 2>/dev/null | grep '^{' > gpurun_out/${TAG}_bench_4k_batch32.json
timeout 600 python bench.py --gpus 2 --dist-backend gloo --batch 1 --no-cpu-baseline --skip-latency --no-copy-ceiling 2>/dev/null | grep '^{' > gpurun_out/${TAG}_bench_two_ranks_one_frame_each_gloo.json
timeout 600 python bench.py --gpus 1 --launcher --no-cpu-baseline --skip-latency --no-other-workloads --no-best-host-config 2>/dev/null | grep '^{' > gpurun_out/${TAG}_bench_one_rank_through_launcher_rccl.json
timeout 600 python bench.py --pool 2 2>/dev/null | grep '^{' > gpurun_out/${TAG}_bench_pool2.json
timeout 600 python bench.py --pool 8 --batch 1 2>/dev/null | grep '^{' > gpurun_out/${TAG}_bench_pool8_one_frame_each.json
timeout 600 python bench.py --shaded --no-cpu-baseline --skip-latency --no-other-workloads 2>/dev/null | grep '^{' > gpurun_out/${TAG}_bench_shaded.json
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r05_bench_*.json")):
    try:
        d = json.load(open(f)); print(f, d["value"], d["ms_per_step"], d["steps"], d["validation"]["mismatching_frames"], d.get("roofline", {}).get("frac"))
    except Exception as e:
        print(f, "ERR", e)
PY
# counter calibration incl. the image-shaped streams
export OUT=$PWD/gpurun_out/pmc_calibration_$TAG; rm -rf $OUT; mkdir -p $OUT
( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o pmc -- $REPO/miniengineao_amd/lib/ubench_fetch > $OUT/fetch.log 2>&1; timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -o pmc -- $REPO/miniengineao_amd/lib/ubench_fetch > $OUT/write.log 2>&1 )
python tools/pmc_calibration.py $OUT > gpurun_out/${TAG}_pmc_calibration.json; find $OUT -name '*kernel_trace.csv' -delete; find $OUT -name '*agent_info.csv' -delete
timeout 120 miniengineao_amd/lib/ubench_fetch > gpurun_out/${TAG}_ubench_fetch_rates.jsonl 2>&1
grep -A6 tile_rows gpurun_out/${TAG}_pmc_calibration.json | grep -E "kernel|over_known"
# one 4K frame per call: L2 requests / hits / misses and vector-memory busy cycles per launch, stored mips vs raw depth
mkdir -p gpurun_out/pmc_${TAG}_from_depth
for grp in "tcc TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA_RDREQ_sum" "tcp TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum"; do
  set -- $grp; name=$1; shift
  ( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $REPO/gpurun_out/pmc_${TAG}_from_depth/$name -o pmc -- python $REPO/tools/from_depth_sweep.py --workloads 4k --frames 1 --iters 20 --rounds 1 > $REPO/gpurun_out/pmc_${TAG}_from_depth/$name.log 2>&1 )
done
python tools/pmc_summary.py gpurun_out/pmc_${TAG}_from_depth > gpurun_out/${TAG}_from_depth_cache_counters.txt 2>&1; grep -E "^==|render|downsample_kernel" gpurun_out/${TAG}_from_depth_cache_counters.txt | cut -c1-400
find gpurun_out/pmc_${TAG}_from_depth -name '*kernel_trace.csv' -delete; find gpurun_out/pmc_${TAG}_from_depth -name '*agent_info.csv' -delete
timeout 1500 python tools/fuzz_gpu.py 1500 500000 > gpurun_out/${TAG}_fuzz_gpu_long.log 2>&1; tail -2 gpurun_out/${TAG}_fuzz_gpu_long.log
MEAO_LIB_PATH=$PWD/miniengineao_amd/lib/variants/libmeao_pair.so timeout 600 python tools/fuzz_gpu.py 300 700000 > gpurun_out/${TAG}_fuzz_gpu_variant_pair.log 2>&1; tail -1 gpurun_out/${TAG}_fuzz_gpu_variant_pair.log
