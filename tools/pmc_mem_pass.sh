#!/bin/bash
# One counter pass on the texture-addresser / L1 path of a small render: how busy the per-lane loads of the tree walk keep it.
# usage: tools/pmc_mem_pass.sh <outdir> [scene] [spp]
set -e
OUT=$(realpath -m "$1"); SCENE=${2:-mega-spheres}; SPP=${3:-8}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$OUT"
export PROBE_REPS=1 TMPDIR=/tmp
cd /tmp
timeout -s KILL 120 rocprofv3 --pmc TA_TA_BUSY_sum TA_BUSY_avr TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVES -d "$OUT/mem" -o p --output-format csv -- python3 "$ROOT/tools/perf_probe.py" "$SCENE" "$SPP" f32 > "$OUT/mem.log" 2>&1
echo done
