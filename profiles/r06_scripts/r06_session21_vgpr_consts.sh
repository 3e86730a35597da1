# Round 6, session 21: blur / bilateral uniforms in VGPRs (variant vconst) against the product, alternating; parity smoke through the variant first.
set -x
mkdir -p gpurun_out
MEAO_LIB_PATH=$PWD/miniengineao_amd/lib/variants/libmeao_vconst.so python tools/parity_probe.py 2>&1 | grep BAD | cut -c1-300
bash profiles/r06_scripts/r06_ab_variants.sh r06s21_pipelined 3 --pipeline --steps 100 --check -- product vconst
bash profiles/r06_scripts/r06_ab_variants.sh r06s21_plain 3 --steps 100 --check -- product vconst
