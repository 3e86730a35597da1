# quick GPU loop: parity tests + bench line (on the GPU box via gpurun)
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline "$@" 2>&1 | grep '^{' > gpurun_out/bench_quick.json
python - <<'PY'
import json
d = json.load(open('gpurun_out/bench_quick.json'))
print('value', d['value'], 'Mpix/s  ms/step', d['ms_per_step'], ' latency_ms', d['single_frame_latency_ms'])
for p in d['roofline']['passes']:
    print('  %-20s %8.2f us/batch  %7.1f GB/s  frac %.3f' % (p['kernel'], p['ms']*1e3, p['GBps'], p['frac']))
print('  whole', d['roofline']['whole_frame'], 'ren+ups', d['roofline']['render_plus_upsample'])
PY
