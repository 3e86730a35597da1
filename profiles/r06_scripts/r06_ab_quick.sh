# (.ab_r05 = `git worktree add .ab_r05 3241d53 && (cd .ab_r05 && python -m miniengineao_amd.build)`: the round-5 HEAD with its own library; removed at the end of the round)
# usage: bash profiles/r06_scripts/r06_ab_quick.sh <tag> [alternations=2]: correctness probe + alternating A/B of bench_passes (r05 worktree vs this tree)
TAG=$1; N=${2:-2}
mkdir -p gpurun_out
python tools/parity_probe.py 2>&1 | grep BAD | cut -c1-200 > gpurun_out/${TAG}_diag.txt
for i in $(seq $N); do
  (cd .ab_r05 && timeout 300 python tools/bench_passes.py --pipeline --steps 100 --tag r05_pipelined) 2>/dev/null | grep '^{' >> gpurun_out/${TAG}_ab.jsonl
  timeout 300 python tools/bench_passes.py --pipeline --steps 100 --check --tag ${TAG}_pipelined 2>/dev/null | grep '^{' >> gpurun_out/${TAG}_ab.jsonl
  (cd .ab_r05 && timeout 300 python tools/bench_passes.py --steps 100 --tag r05_plain) 2>/dev/null | grep '^{' >> gpurun_out/${TAG}_ab.jsonl
  timeout 300 python tools/bench_passes.py --steps 100 --check --tag ${TAG}_plain 2>/dev/null | grep '^{' >> gpurun_out/${TAG}_ab.jsonl
done
cat gpurun_out/${TAG}_diag.txt gpurun_out/${TAG}_ab.jsonl
