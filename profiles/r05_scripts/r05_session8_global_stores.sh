# Round-5 GPU session 8: result stores of the full-resolution pass as global_store (address space spelled out) vs flat_store (what the
# compiler emits for a pointer from the kernel-argument array).  Parity smoke of both, alternating A/B incl. one-frame latency.
set -x
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_variants_gpu.py tests/test_gpu_parity.py -m gpu -q 2>&1 | tail -2
rm -f gpurun_out/r05_ab_global_stores.jsonl
for i in 1 2 3; do for lib in "" miniengineao_amd/lib/variants/libmeao_flatstore.so; do
MEAO_LIB_PATH=${lib:+$PWD/$lib} timeout 300 python bench.py --no-cpu-baseline --no-other-workloads --no-best-host-config --no-copy-ceiling --validate-frames 2 2>/dev/null | grep "^{" | python -c "
import sys,json
d=json.loads(sys.stdin.read()); ps={p['kernel']:round(p['ms']*1e3,1) for p in d['roofline']['passes']}
print(json.dumps({'lib':'${lib##*/}' or 'product','value':d['value'],'ms_per_step':d['ms_per_step'],'passes':ps,'plain_pass_ms':d['plain_launch_sequence']['pass_ms'],'single_frame_us':round(d['single_frame']['direct_back_to_back_ms']*1e3,1),'mismatching':d['validation']['mismatching_frames']}))" >> gpurun_out/r05_ab_global_stores.jsonl
done; done
cat gpurun_out/r05_ab_global_stores.jsonl | cut -c1-460
