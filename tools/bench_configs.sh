#!/bin/bash
# bench.py on the other configurations of the baseline list (the headline one is bench.py's default); one JSON line each.
# Each line carries the f64 figure at top level, the f32 figure under "f32" and, for scenes with a tree of the library's own,
# double precision on that tree under "f64_own_tree".
# usage: tools/bench_configs.sh <out.jsonl>
set -e
OUT=$(realpath -m "$1"); ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT"; : > "$OUT"
run() { timeout -k 10 500 python3 bench.py --steps 2 --warmup 1 "$@" | grep '^{"metric"' >> "$OUT"; }
run --scene random --width 400 --spp 100 --cpu-spp 100
run --scene two-perlin-spheres --spp 256 --cpu-spp 4
run --scene earth --spp 256 --cpu-spp 8
run --scene simple-light --spp 1024 --cpu-spp 8
run --scene cornell-smoke --spp 1024 --cpu-spp 4
run --scene final-scene --width 800 --aspect 1.0 --spp 64 --cpu-spp 2
run --scene random --spp 64 --no-cpu-baseline
run --scene final-scene --spp 64 --no-cpu-baseline
run --scene two-spheres --spp 256 --cpu-spp 8
run --scene mega-spheres --mega-grid 500 --width 3840 --spp 16 --cpu-spp 1
run --scene mega-spheres --mega-grid 500 --width 3840 --spp 16 --cpu-spp 1 --mega-view above
echo done
