# Round-5 GPU session 10: upsample_blend_tall_kernel with the depth window trimmed (22.8 KB, seven workgroups per CU instead of six).
set -x
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_more.py tests/test_large_pipelined.py -m gpu -q -k "blend_with_64x64 or large or pipelined" 2>&1 | tail -2
rm -f gpurun_out/r05_ab_blend_tall_7wg.jsonl
for i in 1 2 3; do for wl in 4k 1080p; do for tall in 0 1; do
timeout 300 python bench.py --workload $wl --blend-tall-min-tiles $tall --no-cpu-baseline --skip-latency --no-other-workloads --no-best-host-config --no-copy-ceiling --validate-frames 2 2>/dev/null | grep "^{" | python -c "
import sys,json
d=json.loads(sys.stdin.read()); ps={p['kernel']:round(p['ms']*1e3,1) for p in d['roofline']['passes']}
print(json.dumps({'workload':'$wl','blend_tall':$tall,'value':d['value'],'ms_per_step':d['ms_per_step'],'passes':ps,'plain_pass_ms':d['plain_launch_sequence']['pass_ms'],'mismatching':d['validation']['mismatching_frames']}))" >> gpurun_out/r05_ab_blend_tall_7wg.jsonl
done; done; done
python - <<'PY'
import json, collections
rows=[json.loads(l) for l in open('gpurun_out/r05_ab_blend_tall_7wg.jsonl')]
agg=collections.defaultdict(list)
for r in rows: agg[(r["workload"], r["blend_tall"])].append(r)
for k,v in sorted(agg.items()):
    print(k, "step", round(sum(x["ms_per_step"] for x in v)/len(v)*1e3,1), "L2L1", round(sum(x["passes"]["upsample_L2_to_L1"] for x in v)/len(v),1), "plain L2L1", round(sum(x["plain_pass_ms"]["upsample_L2_to_L1"] for x in v)/len(v)*1e3,1), [x["mismatching"] for x in v])
PY
