mkdir -p gpurun_out
for i in 1 2; do
for b in 2 4 8 16 32; do
  timeout 300 python tools/bench_passes.py --pipeline --steps $((1600/b)) --batch $b --tag pipelined_b$b 2>/dev/null | grep '^{' >> gpurun_out/r06s19_batch_sweep.jsonl
done
done
cat gpurun_out/r06s19_batch_sweep.jsonl
