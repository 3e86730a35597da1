# round 4, GPU call G: long fuzz on the final kernels + the default bench line on another box + single frames with the side stream
set -x
mkdir -p gpurun_out
T=${1:-r04g}
( time timeout 600 python bench.py --steps 20 --warmup 5 ) 2> gpurun_out/bench_${T}_time.log | grep '^{' > gpurun_out/bench_$T.json
python -c "
import json; d=json.load(open('gpurun_out/bench_$T.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['best_host_config']['value'], d['single_frame'])"
timeout 300 python bench.py --side-stream 4 --no-cpu-baseline --no-other-workloads --no-best-host-config --no-copy-ceiling 2>/dev/null | grep '^{' > gpurun_out/bench_${T}_side4.json
python -c "
import json; d=json.load(open('gpurun_out/bench_${T}_side4.json')); print(d['value'], d['ms_per_step'], d['single_frame'], d['validation']['mismatching_frames'])"
timeout 1500 python tests/fuzz_gpu.py ${2:-2000} 40000 > gpurun_out/fuzz_long_$T.log 2>&1
tail -3 gpurun_out/fuzz_long_$T.log
