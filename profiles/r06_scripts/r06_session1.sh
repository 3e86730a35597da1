# Round 6, session 1: first GPU contact of the LinearDepth-free path (mips-only downsample + raw-depth HiResDB) after the prune.
set -x
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r06s1_smoke.log 2>&1; echo smoke rc=$? >> gpurun_out/r06s1_smoke.log
timeout 1500 python -m pytest tests -m gpu -q -rs > gpurun_out/r06s1_pytest_gpu.log 2>&1; echo pytest rc=$? >> gpurun_out/r06s1_pytest_gpu.log
# alternating A/B on this box: r05 HEAD library + host (.ab_r05, a git worktree of 3241d53) vs this tree
for i in 1 2 3; do
  (cd .ab_r05 && timeout 300 python tools/bench_passes.py --pipeline --steps 100 --tag r05_pipelined) 2>/dev/null | grep '^{' >> gpurun_out/r06s1_ab.jsonl
  timeout 300 python tools/bench_passes.py --pipeline --steps 100 --tag r06_pipelined 2>/dev/null | grep '^{' >> gpurun_out/r06s1_ab.jsonl
  (cd .ab_r05 && timeout 300 python tools/bench_passes.py --steps 100 --tag r05_plain) 2>/dev/null | grep '^{' >> gpurun_out/r06s1_ab.jsonl
  timeout 300 python tools/bench_passes.py --steps 100 --tag r06_plain 2>/dev/null | grep '^{' >> gpurun_out/r06s1_ab.jsonl
done
timeout 300 python tools/bench_passes.py --pipeline --steps 100 --check --tag r06_pipelined_checked 2>/dev/null | grep '^{' >> gpurun_out/r06s1_ab.jsonl
( time timeout 900 python bench.py --steps 20 --warmup 5 ) 2> gpurun_out/r06s1_bench_time.log | grep '^{' > gpurun_out/r06s1_bench_driver_form.json
tail -3 gpurun_out/r06s1_smoke.log; tail -8 gpurun_out/r06s1_pytest_gpu.log; cat gpurun_out/r06s1_ab.jsonl; cut -c1-600 gpurun_out/r06s1_bench_driver_form.json
