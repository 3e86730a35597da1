# Round 6, session 16: the from-raw LoResDB window of the full-resolution pass (MEAO_X_LOWDEPTH_FROM_RAW) -- parity first, then
# product vs variant `lowbuf` alternating on this box (pipelined and plain launch sequence).
set -x
mkdir -p gpurun_out
python tools/parity_probe.py 2>&1 | grep BAD | cut -c1-300 > gpurun_out/r06s16_diag.txt; cat gpurun_out/r06s16_diag.txt
timeout 1500 python -m pytest tests -m gpu -q -x -rs > gpurun_out/r06s16_pytest.log 2>&1; echo pytest rc=$? >> gpurun_out/r06s16_pytest.log; tail -5 gpurun_out/r06s16_pytest.log
bash profiles/r06_scripts/r06_ab_variants.sh r06s16_pipelined 3 --pipeline --steps 100 --check -- lowbuf product
bash profiles/r06_scripts/r06_ab_variants.sh r06s16_plain 3 --steps 100 --check -- lowbuf product
