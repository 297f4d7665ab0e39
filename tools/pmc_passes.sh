#!/bin/bash
# Counter passes over one render of the headline frame at 32 samples per pixel (one pass of the sample loop,
# 1 generation + 50 bounce dispatches).  Each pass is its own rocprofv3 run: counters only, no API tracing.
# usage: tools/pmc_passes.sh <outdir> [scene] [precision]
set -e
OUT=$(realpath -m "$1"); SCENE=${2:-cornell-box}; PREC=${3:-f32}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$OUT"
export PROBE_REPS=1 TMPDIR=/tmp
cd /tmp
pass() {
  name=$1; shift
  timeout -s KILL 120 rocprofv3 --pmc "$@" -d "$OUT/$name" -o p --output-format csv -- python3 "$ROOT/tools/perf_probe.py" "$SCENE" 32 "$PREC" > "$OUT/$name.log" 2>&1
}
pass a SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_BRANCH
pass b SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_THREAD_CYCLES_VALU
pass c SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_SALU SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32
pass d FETCH_SIZE GRBM_GUI_ACTIVE
pass e WRITE_SIZE
echo done
