mkdir -p gpurun_out
for wl in 4k 1080p; do
  timeout 600 python tools/single_frame_sweep.py --workload $wl > gpurun_out/r06_single_frame_sweep_$wl.jsonl 2>/dev/null
  timeout 600 python tools/single_frame_sweep.py --workload $wl --pipelined > gpurun_out/r06_single_frame_sweep_${wl}_pipelined.jsonl 2>/dev/null
done
head -4 gpurun_out/r06_single_frame_sweep_*.jsonl
