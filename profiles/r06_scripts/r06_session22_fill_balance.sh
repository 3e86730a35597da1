# Round 6, session 22: which waves of a from-raw tile take the partial second AO round / the apron items (MEAO_X_FILL_BALANCE 0 / 1 / 2).
set -x
mkdir -p gpurun_out
for v in fill1 fill2; do MEAO_LIB_PATH=$PWD/miniengineao_amd/lib/variants/libmeao_$v.so python tools/parity_probe.py 2>&1 | grep BAD | cut -c1-200; done
bash profiles/r06_scripts/r06_ab_variants.sh r06s22_pipelined 3 --pipeline --steps 100 --check -- product fill1 fill2
bash profiles/r06_scripts/r06_ab_variants.sh r06s22_plain 3 --steps 100 --check -- product fill1 fill2
