#!/bin/bash
# A/B of two builds of the library on one box: tools/ab_probe.sh <build dir A> <build dir B> [scene] [spp]; prints Grays/s of both
# precisions, two rounds each, alternating.
A=$1; B=$2; SCENE=${3:-cornell-box}; SPP=${4:-256}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
for round in 1 2; do
  for d in "$A" "$B"; do
    for prec in f64 f32; do
      echo -n "$(basename $d) $prec: "
      RTAMD_BUILD_DIR=$d PROBE_REPS=3 timeout -k 10 200 python3 "$ROOT/tools/perf_probe.py" $SCENE $SPP $prec 2>/dev/null | tail -n 1 | sed 's/.*-> //'
    done
  done
done
