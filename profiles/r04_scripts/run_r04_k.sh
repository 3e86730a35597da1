# round 4, GPU call K: full suite at HEAD, default line, rocprofv3 trace of the side-stream step
set -x
mkdir -p gpurun_out
T=r04k
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/pytest_$T.log 2>&1; echo rc=$? >> gpurun_out/pytest_$T.log
tail -4 gpurun_out/pytest_$T.log
( time timeout 600 python bench.py --steps 20 --warmup 5 ) 2> gpurun_out/bench_${T}_time.log | grep '^{' > gpurun_out/bench_$T.json
python -c "
import json; d=json.load(open('gpurun_out/bench_$T.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['best_host_config']['value'], d['single_frame'])"
REPO=$PWD
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/prof_${T}_side4 -o trace -- python $REPO/bench.py --side-stream 4 --no-cpu-baseline --skip-latency --no-other-workloads --no-copy-ceiling --no-best-host-config --no-pipeline-plain-leg > $REPO/gpurun_out/prof_${T}_side4.log 2>&1 || true
cd $REPO
f=$(find gpurun_out/prof_${T}_side4 -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && head -12 "$f"
python tests/rocprof_timed_region.py $(find gpurun_out/prof_${T}_side4 -name '*kernel_trace.csv' | head -1) 30 2>/dev/null | head -12
find gpurun_out/prof_${T}_side4 -name '*kernel_trace.csv' -size +20M -delete
