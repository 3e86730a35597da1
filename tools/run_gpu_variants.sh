# variants on the GPU: fuzz sweep (incl. the 8f#4 variants) + bench lines of the variants
mkdir -p gpurun_out
timeout 900 python tools/fuzz_gpu.py ${1:-300} ${2:-9000} 2>&1 | tail -5
for v in "" "--hq-levels 4" "--hq-levels 2" "--exhaustive" "--hq-levels 4 --exhaustive"; do
  tag=$(echo "base $v" | tr -d ' -')
  timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --skip-latency $v 2>&1 | grep '^{' > gpurun_out/variant_$tag.json
  python - gpurun_out/variant_$tag.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print(sys.argv[1], 'value', d['value'], 'ms/step', d['ms_per_step'], ' '.join('%s=%.1f' % (p['kernel'].replace('upsample_', 'u'), p['ms'] * 1e3) for p in d['roofline']['passes']))
PY
done
