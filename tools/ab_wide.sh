#!/bin/bash
# Four-wide walk against the threaded list (RTAMD_NO_WIDE=1) on the per-lane-walk scenes that live in memory: Grays/s of the
# million-sphere field at 4K and of final-scene at 1080p, single precision, for the in-tree build and every _build_* variant beside it.
ROOT=$(cd "$(dirname "$0")/.." && pwd)
probe() { timeout -k 10 200 python3 "$ROOT/tools/perf_probe.py" "$@" 2>&1 | grep -E "Grays|four-wide tree" | sed 's/.*rays \(.*\)/\1/'; }
export PROBE_REPS=2
echo "== list: mega-spheres 4K x16 f32"; RTAMD_NO_WIDE=1 PROBE_MEGA=500 probe mega-spheres 16 ${PREC:-f32} 0 3840 || exit 1
echo "== list: final-scene 1080p x64 f32"; RTAMD_NO_WIDE=1 probe final-scene 64 ${PREC:-f32} || exit 1
echo "== list: random 1080p x64 f32"; RTAMD_NO_WIDE=1 probe random 64 ${PREC:-f32} || exit 1
for d in "$ROOT"/rust-raytracing_amd/_build "$ROOT"/rust-raytracing_amd/_build_*; do
  [ -d "$d" ] || continue
  echo "== $(basename $d) wide: mega-spheres 4K x16"; RTAMD_BUILD_DIR=$d RTAMD_TIMING=1 PROBE_MEGA=500 probe mega-spheres 16 ${PREC:-f32} 0 3840 || exit 1
  echo "== $(basename $d) wide: final-scene 1080p x64"; RTAMD_BUILD_DIR=$d probe final-scene 64 ${PREC:-f32} || exit 1
  echo "== $(basename $d) wide: random 1080p x64"; RTAMD_BUILD_DIR=$d probe random 64 ${PREC:-f32} || exit 1
done
