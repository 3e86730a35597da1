# Round-5 GPU session 9: the L2 -> L1 blend pass with 64 x 64 tiles (upsample_blend_tall_kernel) for large batches.  Parity, then an
# alternating A/B on the three workloads.
set -x
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_more.py -m gpu -q -k "blend_with_64x64" 2>&1 | tail -2
rm -f gpurun_out/r05_ab_blend_tall.jsonl
for i in 1 2 3; do for wl in 4k 1080p 8k; do for tall in 0 1; do
timeout 300 python bench.py --workload $wl --blend-tall-min-tiles $tall --no-cpu-baseline --skip-latency --no-other-workloads --no-best-host-config --no-copy-ceiling --validate-frames 2 2>/dev/null | grep "^{" | python -c "
import sys,json
d=json.loads(sys.stdin.read()); ps={p['kernel']:round(p['ms']*1e3,1) for p in d['roofline']['passes']}
print(json.dumps({'workload':'$wl','blend_tall':$tall,'value':d['value'],'ms_per_step':d['ms_per_step'],'passes':ps,'plain_pass_ms':d['plain_launch_sequence']['pass_ms'],'mismatching':d['validation']['mismatching_frames']}))" >> gpurun_out/r05_ab_blend_tall.jsonl
done; done; done
cat gpurun_out/r05_ab_blend_tall.jsonl | cut -c1-400
