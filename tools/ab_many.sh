#!/bin/bash
# Several builds of the library on one box: tools/ab_many.sh "<scene> <spp>" <build dir> ...; Grays/s of both precisions, two rounds.
ARGS=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
for round in 1 2; do
  for d in "$@"; do
    for prec in f64 f32; do
      echo -n "$(basename $d) $prec: "
      RTAMD_BUILD_DIR=$d PROBE_REPS=3 timeout -k 10 200 python3 "$ROOT/tools/perf_probe.py" $ARGS $prec 2>/dev/null | tail -n 1 | sed 's/.*-> //'
    done
  done
done
