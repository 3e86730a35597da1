# usage: bash tools/run_rocprof.sh <tag> [bench args...]   (on the GPU box, via gpurun)
set -x
TAG=${1:-r01}; shift
REPO=$(pwd)
mkdir -p $REPO/gpurun_out/prof_$TAG
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/prof_$TAG -o trace -- python $REPO/bench.py --no-cpu-baseline --skip-latency --no-other-workloads --no-copy-ceiling --no-best-host-config "$@" > $REPO/gpurun_out/prof_$TAG/bench_under_rocprof.log 2>&1
echo rc=$? >> $REPO/gpurun_out/prof_$TAG/bench_under_rocprof.log
cd $REPO
find gpurun_out/prof_$TAG -name '*stats*' | head; 
f=$(find gpurun_out/prof_$TAG -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && head -20 "$f"
# keep the merge small: drop the raw per-dispatch trace if it is large
find gpurun_out/prof_$TAG -name '*kernel_trace.csv' -size +20M -delete
tail -2 gpurun_out/prof_$TAG/bench_under_rocprof.log
