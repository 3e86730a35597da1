set -x
REPO=$(pwd); OUT=$REPO/gpurun_out/pmc_r02d; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() { name=$1; shift; rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -o pmc -- python $REPO/bench.py --no-cpu-baseline --skip-latency --no-copy-ceiling --no-best-host-config --steps 3 --warmup 1 > $OUT/$name.log 2>&1; echo "$name rc=$?"; }
run sq3 SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT
run sq4 SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU2 SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES SQ_INSTS_VALU_TRANS_F32 SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_INST_CYCLES_SALU
cd $REPO; find $OUT -name '*kernel_trace.csv' -delete; find $OUT -name '*agent_info.csv' -delete
python tools/pmc_summary.py $OUT
