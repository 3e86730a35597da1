# round 4, GPU call L: rocprofv3 kernel trace of the side-stream step (shows the side kernel overlapping the launches of the call)
set -x
mkdir -p gpurun_out
T=r04l
REPO=$PWD
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/prof_${T}_side4 -o trace -- python $REPO/bench.py --side-stream 4 --no-cpu-baseline --skip-latency --no-other-workloads --no-copy-ceiling --no-best-host-config > $REPO/gpurun_out/prof_${T}_side4.log 2>&1
cd $REPO
f=$(find gpurun_out/prof_${T}_side4 -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && head -12 "$f" | cut -c1-200
python tests/rocprof_timed_region.py $(find gpurun_out/prof_${T}_side4 -name '*kernel_trace.csv' | head -1) 30 | head -12
python - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/prof_r04l_side4/**/*kernel_trace.csv', recursive=True)[0]
rows = [r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# one pipelined step near the end: the last side kernel and every launch that overlaps it
side = [r for r in rows if 'downsample_side_kernel' in r['Kernel_Name']][-3]
s0, s1 = int(side['Start_Timestamp']), int(side['End_Timestamp'])
print('side kernel', 0.0, round((s1 - s0) / 1e3, 1), 'us')
for r in rows:
    a, b = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    if b > s0 - 50000 and a < s1 + 250000 and 'side' not in r['Kernel_Name']:
        print(r['Kernel_Name'].split('(')[0][-60:], round((a - s0) / 1e3, 1), round((b - s0) / 1e3, 1))
PY
find gpurun_out/prof_${T}_side4 -name '*kernel_trace.csv' -delete
