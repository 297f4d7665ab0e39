#!/bin/bash
# One scene and precision on several builds of the library: tools/ab_scene.sh "<perf_probe args>" <build dir> ...; two rounds.
ARGS=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
for round in 1 2; do
  for d in "$@"; do
    echo -n "$(basename $d) [$ARGS]: "
    RTAMD_BUILD_DIR=$d PROBE_REPS=2 timeout -k 10 200 python3 "$ROOT/tools/perf_probe.py" $ARGS 2>/dev/null | tail -n 1 | sed 's/.*-> //'
  done
done
