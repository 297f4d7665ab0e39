#!/bin/bash
# Per-lane-walk scenes on several builds of the library: tools/ab_stepwise.sh <build dir> ...; prints Grays/s for `random` at 1080p
# and the million-sphere field at 4K, both precisions.
ROOT=$(cd "$(dirname "$0")/.." && pwd)
for d in "$@"; do
  for prec in f32 f64; do
    echo -n "$(basename $d) random $prec: "
    RTAMD_BUILD_DIR=$d PROBE_REPS=2 timeout -k 10 200 python3 "$ROOT/tools/perf_probe.py" random 64 $prec 2>/dev/null | tail -n 1 | sed 's/.*-> //'
    echo -n "$(basename $d) mega $prec: "
    RTAMD_BUILD_DIR=$d PROBE_REPS=2 PROBE_MEGA=500 timeout -k 10 200 python3 "$ROOT/tools/perf_probe.py" mega-spheres 16 $prec 0 3840 2>/dev/null | tail -n 1 | sed 's/.*-> //'
  done
done
