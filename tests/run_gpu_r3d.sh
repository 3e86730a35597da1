set -x
TAG=${1:-r03d}
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_variants_gpu.py -m gpu -q 2>&1 | tail -5
bash tests/run_gpu_ab3.sh $TAG 2
