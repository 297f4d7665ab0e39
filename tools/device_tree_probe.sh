#!/bin/bash
# The million-sphere frame with the tree built on the device and with the host builder (RTAMD_NO_GPU_BUILD=1): upload stages and
# frame rates side by side, both cameras, both precisions.
# usage: tools/device_tree_probe.sh [spp]
SPP=${1:-16}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT"
export PROBE_MEGA=500 RTAMD_TIMING=1
for view in 0 1; do
  for host in "" 1; do
    for prec in f32 f64; do
      echo "== view $view, ${host:+host }${host:-device }builder, $prec"
      if [ -n "$host" ]; then export RTAMD_NO_GPU_BUILD=1; else unset RTAMD_NO_GPU_BUILD; fi
      PROBE_MEGA_VIEW=$view timeout -k 10 200 python3 tools/perf_probe.py mega-spheres $SPP $prec 0 3840 2>&1 | grep -E "own tree|device-built|four-wide|upload|Grays" || exit 1
    done
  done
done
