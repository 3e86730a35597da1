# usage: bash profiles/r06_scripts/r06_ab_variants.sh <tag> <alternations> <bench_passes args...> -- variant names (product = the product library)
TAG=$1; N=$2; shift 2
ARGS=""; while [ "$1" != "--" ]; do ARGS="$ARGS $1"; shift; done; shift
mkdir -p gpurun_out
for i in $(seq $N); do
  for v in "$@"; do
    if [ $v = product ]; then LIBP=$PWD/miniengineao_amd/lib/libmeao_hip.so; else LIBP=$PWD/miniengineao_amd/lib/variants/libmeao_$v.so; fi
    MEAO_LIB_PATH=$LIBP timeout 300 python tools/bench_passes.py $ARGS --tag $v 2>/dev/null | grep '^{' >> gpurun_out/${TAG}.jsonl
  done
done
cat gpurun_out/${TAG}.jsonl
