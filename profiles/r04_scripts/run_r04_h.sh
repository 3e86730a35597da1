# round 4, GPU call H: single-frame launch-structure sweep with the round-4 blend kernels; fuzz timing
set -x
mkdir -p gpurun_out
T=r04h
timeout 600 python tools/single_frame_sweep.py --workload 4k > gpurun_out/single_frame_sweep_4k_$T.jsonl 2>/dev/null
head -6 gpurun_out/single_frame_sweep_4k_$T.jsonl
timeout 600 python tools/single_frame_sweep.py --workload 4k --pipelined > gpurun_out/single_frame_sweep_4k_pipelined_$T.jsonl 2>/dev/null
head -4 gpurun_out/single_frame_sweep_4k_pipelined_$T.jsonl
timeout 600 python tools/single_frame_sweep.py --workload 1080p > gpurun_out/single_frame_sweep_1080p_$T.jsonl 2>/dev/null
head -4 gpurun_out/single_frame_sweep_1080p_$T.jsonl
( time timeout 900 python tests/fuzz_gpu.py 300 90000 ) > gpurun_out/fuzz_timing_$T.log 2>&1
tail -5 gpurun_out/fuzz_timing_$T.log
