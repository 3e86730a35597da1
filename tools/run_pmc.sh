# usage: [PMC_GROUPS="sq1 sq5"] bash tools/run_pmc.sh <tag> [bench args...]   -- PMC passes, one counter group per run
set -x
TAG=${1:-r01}; shift
REPO=$(pwd)
export OUT=$REPO/gpurun_out/pmc_$TAG
mkdir -p $OUT
python -m miniengineao_amd.codehash > $OUT/code_sha256.txt      # the device code these counters belong to (profiles/pmc_traffic.json "_code_sha256")
cd /tmp && export TMPDIR=/tmp
timeout 120 rocprofv3 -L > $OUT/counters_available.txt 2>&1
run() { name=$1; shift; case " ${PMC_GROUPS:-sq1 sq2 sq3 sq4 sq5 fetch write} " in *" $name "*) ;; *) return;; esac; timeout 400 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -o pmc -- python $REPO/bench.py --no-cpu-baseline --skip-latency --no-other-workloads --no-copy-ceiling --no-best-host-config --validate-frames 0 --min-time-ms 0 --steps 3 --warmup 1 $BENCH_ARGS > $OUT/$name.log 2>&1; echo "$name rc=$?"; }
BENCH_ARGS="$*"
run sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY
run sq2 SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE
run sq3 SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT
run sq4 SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU2 SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES SQ_INSTS_VALU_TRANS_F32 SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_INST_CYCLES_SALU
run sq5 SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64
run fetch FETCH_SIZE
run write WRITE_SIZE
cd $REPO
python tools/pmc_summary.py $OUT
find $OUT -name '*kernel_trace.csv' -delete; find $OUT -name '*agent_info.csv' -delete
