# Round 6, session 20: cheaper addressing in the upsample tiles (no clamps for tiles inside the frame, i / 10 as multiply + shift, the apron
# item computed once) -- parity, then `prev` (the library of the committed evidence) against the product, alternating.
set -x
mkdir -p gpurun_out
python tools/parity_probe.py 2>&1 | grep BAD | cut -c1-300
timeout 900 python -m pytest tests/test_from_raw_window.py tests/test_gpu_parity.py tests/test_reference_goldens.py -m gpu -q -x 2>&1 | tail -3
bash profiles/r06_scripts/r06_ab_variants.sh r06s20_pipelined 3 --pipeline --steps 100 --check -- prev product
bash profiles/r06_scripts/r06_ab_variants.sh r06s20_plain 2 --steps 100 --check -- prev product
