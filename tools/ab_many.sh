#!/bin/bash
# Several builds of the library on one box: tools/ab_many.sh <scene> <spp> <precision> <build dir>...; prints Grays/s of each, two rounds.
SCENE=$1; SPP=$2; PREC=$3; shift; shift; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
for round in 1 2; do
  for d in "$@"; do
    echo -n "$(basename $d) $SCENE $PREC: "
    RTAMD_BUILD_DIR=$d PROBE_REPS=3 timeout -k 10 200 python3 "$ROOT/tools/perf_probe.py" $SCENE $SPP $PREC 2>/dev/null | tail -n 1 | sed 's/.*-> //'
  done
done
