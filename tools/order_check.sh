#!/bin/bash
# Child order of the library's own tree (run through gpurun): camera-first against the builder's order.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT"
for sc in random mega-spheres final-scene; do
  echo "$sc camera-first:  $(timeout -k 5 100 python3 tools/perf_probe.py $sc 64 f32 | grep -o 'rays.*')"
  echo "$sc builder order: $(RTAMD_NO_CAMERA_ORDER=1 timeout -k 5 100 python3 tools/perf_probe.py $sc 64 f32 | grep -o 'rays.*')"
done
