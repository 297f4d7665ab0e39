#!/bin/bash
# Profiles of the bench command under one tag, written under gpurun_out/profile_<tag>/:
#   trace/   rocprofv3 --kernel-trace --stats of `python3 bench.py` (the driver's command, CPU baseline included)
#   pmc_<precision>_*/   counter passes (one rocprofv3 run each, counters only) of
#            `python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --precision <precision>`, for f64 and f32
# usage: tools/profile_round.sh <tag> [extra bench args...]
set -e
TAG=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT="$ROOT/gpurun_out/profile_$TAG"
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
timeout -s KILL ${TRACE_TIMEOUT:-400} rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o bench --output-format csv -- python3 "$ROOT/bench.py" "$@" > "$OUT/trace.log" 2>&1
grep '^{"metric"' "$OUT/trace.log" > "$OUT/bench_line.json" || true
pass() {
  prec=$1; name=$2; shift; shift
  timeout -s KILL ${PASS_TIMEOUT:-180} rocprofv3 --pmc "$@" -d "$OUT/pmc_${prec}_$name" -o p --output-format csv -- python3 "$ROOT/bench.py" --steps 1 --warmup 0 --no-cpu-baseline --precision $prec $BENCH_ARGS > "$OUT/pmc_${prec}_$name.log" 2>&1
}
BENCH_ARGS="$*"
for prec in ${PROFILE_PRECISIONS:-f64 f32}; do
  pass $prec fetch FETCH_SIZE GRBM_GUI_ACTIVE
  pass $prec write WRITE_SIZE
  pass $prec sq_a SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_BRANCH
  pass $prec sq_b SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_THREAD_CYCLES_VALU
  pass $prec sq_c SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_CVT
  pass $prec mem TA_TA_BUSY_sum TA_BUSY_avr TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE
  pass $prec tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
  pass $prec sq_d SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT64 SQ_INST_CYCLES_SALU SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES
done
echo "profile_round $TAG done"
