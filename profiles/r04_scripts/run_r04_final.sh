# round 4, last GPU call: the evidence round at HEAD, then 3000 random configurations through the final kernels
bash tests/run_gpu_round.sh r04
( time timeout 1200 python tests/fuzz_gpu.py 3000 120000 ) > gpurun_out/fuzz_long_r04.log 2>&1
tail -5 gpurun_out/fuzz_long_r04.log
