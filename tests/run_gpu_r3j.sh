TAG=${1:-r03j}
mkdir -p gpurun_out /tmp/hold
mv miniengineao_amd/lib/variants/libmeao_p3clocks.so /tmp/hold/
timeout 900 python -m pytest tests/test_variants_gpu.py -m gpu -q 2>&1 | tail -3
bash tests/run_gpu_ab3.sh $TAG 2
MEAO_LIB_PATH=/tmp/hold/libmeao_p3clocks.so python tools/wg_log.py 2>&1 | grep -v amdgpu.ids | head -12
