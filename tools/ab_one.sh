# usage: ab_one.sh "<variant> [variant ...]" [rounds]     product vs variant libraries, plain and pipelined per-pass times
VS=$1; R=${2:-3}
for r in $(seq $R); do for lib in product $VS; do
  if [ $lib = product ]; then unset MEAO_LIB_PATH; else export MEAO_LIB_PATH=$PWD/miniengineao_amd/lib/variants/libmeao_$lib.so; fi
  for mode in "" "--pipeline"; do
  timeout 200 python tools/bench_passes.py --check $mode 2>/dev/null | python -c "
import sys,json
r=json.loads(sys.stdin.readline())
print('$lib', '$mode' or 'plain', r['step_us'], r['render'], r.get('upsample_L3_to_L2'), r['upsample_L2_to_L1'], r['upsample_L1_to_L0'], r['ok'])"
  done; done; done
