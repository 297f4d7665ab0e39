#!/bin/bash
# The default grab policy on frames of different lengths (run through gpurun): one bench.py line per configuration.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT"
one() {
  line=$(timeout -k 10 200 python3 bench.py "$@" --no-cpu-baseline | grep '^{"metric"')
  echo "$* -> $(echo "$line" | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print("%.0f Mrays/s  %.3f ms/step" % (d["value"], d["ms_per_step"]))')"
}
one --spp 2 --steps 20 --warmup 3
one --spp 16 --steps 20 --warmup 3
one --spp 128 --steps 10 --warmup 2
one --spp 512 --steps 3 --warmup 1
one --scene random --width 400 --spp 100 --steps 20 --warmup 3
