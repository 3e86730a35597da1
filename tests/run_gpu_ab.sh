# A/B on ONE box: bench with the in-tree lib, then rebuild with the given hipcc defines and bench again, twice each
# usage: bash tests/run_gpu_ab.sh "-DMEAO_UPS_TILE_H=32" [bench args]
mkdir -p gpurun_out
DEF="$1"; shift
show() { python - "$1" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print(sys.argv[1], 'value', d['value'], ' '.join('%s=%.1f' % (p['kernel'].replace('upsample_', 'u'), p['ms'] * 1e3) for p in d['roofline']['passes']))
PY
}
run() { timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --skip-latency "${@:2}" 2>&1 | grep '^{' > gpurun_out/ab_$1.json; show gpurun_out/ab_$1.json; }
run A1 "$@"
python -c "from miniengineao_amd import build; build.build_lib(force=True, extra_flags=tuple('$DEF'.split()))"
run B1 "$@"
python -c "from miniengineao_amd import build; build.build_lib(force=True)"
run A2 "$@"
python -c "from miniengineao_amd import build; build.build_lib(force=True, extra_flags=tuple('$DEF'.split()))"
run B2 "$@"
