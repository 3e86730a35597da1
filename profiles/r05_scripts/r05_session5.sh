# Round-5 GPU session 5: part of the next step's downsample pass as extra workgroups of the render launch (one stream, no events):
# MEAO_DEBUG_DS_SHARE_IN_RENDER.  Parity first, then an alternating A/B over the share.
set -x
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_more.py -m gpu -q -k "carried_downsample_split" > gpurun_out/r05_pytest_ds_share_in_render.log 2>&1; tail -3 gpurun_out/r05_pytest_ds_share_in_render.log
rm -f gpurun_out/r05_ab_ds_share_in_render.jsonl
for i in 1 2 3; do for share in 0 20 30 40 50 60; do
timeout 300 python bench.py --ds-share-in-render $share --no-cpu-baseline --skip-latency --no-other-workloads --no-best-host-config --no-copy-ceiling --validate-frames 2 2>/dev/null | grep "^{" | python -c "
import sys,json
d=json.loads(sys.stdin.read()); ps={p['kernel']:round(p['ms']*1e3,1) for p in d['roofline']['passes']}
print(json.dumps({'ds_share_in_render':$share,'value':d['value'],'ms_per_step':d['ms_per_step'],'passes':ps,'mismatching':d['validation']['mismatching_frames'],'frames_vs_oracle':d['validation']['frames_vs_oracle']}))" >> gpurun_out/r05_ab_ds_share_in_render.jsonl
done; done
cat gpurun_out/r05_ab_ds_share_in_render.jsonl
