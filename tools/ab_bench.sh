# A/B of the default bench line: product library vs every variant library, alternating, two rounds
for i in 1 2; do for lib in "" miniengineao_amd/lib/variants/libmeao_*.so; do
MEAO_LIB_PATH=${lib:+$PWD/$lib} timeout 300 python bench.py --no-cpu-baseline ${AB_BENCH_ARGS} 2>/dev/null | grep "^{" | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('${lib##*/}' or 'product', d['value'], d['ms_per_step'], 'plain', d['plain_launch_sequence']['value'], json.dumps(d['single_frame'])[:140])"
done; done
