# A/B of the shaded line (depth in -> shaded frame out): every variant library, alternating, two rounds
for i in 1 2; do for lib in miniengineao_amd/lib/variants/libmeao_*.so; do
MEAO_LIB_PATH=$PWD/$lib timeout 300 python bench.py --shaded --no-cpu-baseline --skip-latency 2>/dev/null | grep "^{" | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('${lib##*/}', d['value'], json.dumps(d['depth_in_to_shaded_frame_out'])[:230])"
done; done
