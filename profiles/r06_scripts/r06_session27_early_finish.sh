# Round 6, session 27: the carried downsample tile finished (linearized and stored) between the two halves of the bilateral phase instead of
# at the end of the workgroup (variant ef1: its loads issued with the tile's own; ef2: loads where they were) -- parity of the pipelined path,
# then product / ef1 / ef2 alternating.
set -x
mkdir -p gpurun_out
for v in ef1 ef2; do
  MEAO_LIB_PATH=$PWD/miniengineao_amd/lib/variants/libmeao_$v.so timeout 600 python -m pytest tests/test_large_pipelined.py tests/test_from_raw_window.py -m gpu -q -x -k "pipelined" 2>&1 | tail -2
done
bash profiles/r06_scripts/r06_ab_variants.sh r06s27_pipelined 3 --pipeline --steps 100 --check -- product ef1 ef2
