# Round-5 GPU session 1: parity (suite + new fixtures), default bench lines, one-frame-per-call forms, paired-reciprocal A/B, PMC calibration.
set -x
TAG=r05
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${TAG}_smoke.log 2>&1; echo smoke rc=$? >> gpurun_out/${TAG}_smoke.log
timeout 1200 python -m pytest tests -m gpu -q --maxfail=40 > gpurun_out/${TAG}_pytest_gpu.log 2>&1; echo pytest rc=$? >> gpurun_out/${TAG}_pytest_gpu.log
tail -5 gpurun_out/${TAG}_pytest_gpu.log
( time timeout 600 python bench.py --steps 20 --warmup 5 ) 2> gpurun_out/${TAG}_bench_driver_form_time.log | grep '^{' > gpurun_out/${TAG}_bench_driver_form.json
timeout 600 python bench.py 2> gpurun_out/${TAG}_bench_4k.err | grep '^{' > gpurun_out/${TAG}_bench_4k.json
timeout 600 python bench.py --workload 1080p --no-other-workloads --no-cpu-baseline 2>/dev/null | grep '^{' > gpurun_out/${TAG}_bench_1080p.json
timeout 600 python tools/from_depth_sweep.py > gpurun_out/${TAG}_from_depth_sweep.jsonl 2> gpurun_out/${TAG}_from_depth_sweep.err
# A/B, alternating, 3 rounds: product (three reciprocals) | nopair (five) | the round-4 library as committed at b7bd9dc
for i in 1 2 3; do for lib in "" miniengineao_amd/lib/variants/libmeao_nopair.so miniengineao_amd/lib/ab/libmeao_r04.so; do
MEAO_LIB_PATH=${lib:+$PWD/$lib} timeout 300 python bench.py --no-cpu-baseline --skip-latency --no-other-workloads --no-best-host-config --no-copy-ceiling --validate-frames 2 2>/dev/null | grep "^{" | python -c "
import sys,json
d=json.loads(sys.stdin.read()); ps={p['kernel']:round(p['ms']*1e3,1) for p in d['roofline']['passes']}
print(json.dumps({'lib':'${lib##*/}' or 'product','value':d['value'],'ms_per_step':d['ms_per_step'],'passes':ps,'plain':d['plain_launch_sequence']['value'],'plain_pass_ms':d['plain_launch_sequence']['pass_ms'],'mismatching':d['validation']['mismatching_frames']}))" >> gpurun_out/${TAG}_ab_pair_rcp.jsonl
done; done
cat gpurun_out/${TAG}_ab_pair_rcp.jsonl | cut -c1-330
# PMC calibration on streams of known size
export OUT=$PWD/gpurun_out/pmc_calibration_$TAG; mkdir -p $OUT; REPO=$PWD
( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o pmc -- $REPO/miniengineao_amd/lib/ubench_fetch > $OUT/fetch.log 2>&1; timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -o pmc -- $REPO/miniengineao_amd/lib/ubench_fetch > $OUT/write.log 2>&1 )
python tools/pmc_calibration.py $OUT > gpurun_out/${TAG}_pmc_calibration.json; find $OUT -name '*kernel_trace.csv' -delete; find $OUT -name '*agent_info.csv' -delete
timeout 120 miniengineao_amd/lib/ubench_fetch > gpurun_out/${TAG}_ubench_fetch_rates.jsonl 2>&1
cut -c1-600 gpurun_out/${TAG}_bench_4k.json; grep -o '"single_frame": {[^}]*}[^}]*}' gpurun_out/${TAG}_bench_4k.json gpurun_out/${TAG}_bench_1080p.json; cat gpurun_out/${TAG}_from_depth_sweep.jsonl; head -c 1500 gpurun_out/${TAG}_pmc_calibration.json
