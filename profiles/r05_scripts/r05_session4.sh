# Round-5 GPU session 4: the announced downsample pass split between a side-stream co-runner of render (first s tenths of the frames,
# released at the start of the call) and the last kernel (the rest): MEAO_DEBUG_DS_SIDE_STREAM 5s004.  A/B, alternating, 3 rounds.
set -x
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_more.py -m gpu -q -k "side_stream or captured_launch" > gpurun_out/r05_pytest_side_stream_split.log 2>&1; tail -3 gpurun_out/r05_pytest_side_stream_split.log
for i in 1 2 3; do for mode in 0 4 52004 53004 54004 55004 56004 54014; do
timeout 300 python bench.py --side-stream $mode --no-cpu-baseline --skip-latency --no-other-workloads --no-best-host-config --no-copy-ceiling --validate-frames 2 2>/dev/null | grep "^{" | python -c "
import sys,json
d=json.loads(sys.stdin.read()); ps={p['kernel']:round(p['ms']*1e3,1) for p in d['roofline']['passes']}
print(json.dumps({'side_stream_mode':$mode,'value':d['value'],'ms_per_step':d['ms_per_step'],'passes':ps,'mismatching':d['validation']['mismatching_frames'],'frames_vs_oracle':d['validation']['frames_vs_oracle']}))" >> gpurun_out/r05_ab_side_stream_split.jsonl
done; done
cat gpurun_out/r05_ab_side_stream_split.jsonl
