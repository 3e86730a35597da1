mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_more.py tests/test_large_pipelined.py tests/test_reference_goldens.py tests/test_hostile_depth.py -m gpu -q -x -k "pipelined or hostile" 2>&1 | tail -8 > gpurun_out/r06s4_pytest.log
for i in 1 2 3; do
  timeout 300 python tools/bench_passes.py --pipeline --steps 100 --check --tag in_render 2>/dev/null | grep '^{' >> gpurun_out/r06s4_ab.jsonl
  timeout 300 python tools/bench_passes.py --pipeline --steps 100 --check --debug-set NEXT_DOWNSAMPLE_CARRIER=1 --tag in_final 2>/dev/null | grep '^{' >> gpurun_out/r06s4_ab.jsonl
  timeout 300 python tools/bench_passes.py --pipeline --steps 100 --check --debug-set NEXT_DOWNSAMPLE_CARRIER=2 --tag own_launch 2>/dev/null | grep '^{' >> gpurun_out/r06s4_ab.jsonl
done
cat gpurun_out/r06s4_pytest.log gpurun_out/r06s4_ab.jsonl
