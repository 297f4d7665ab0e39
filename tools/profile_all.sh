#!/bin/bash
# The profile set kept under profiles/ in one go: tools/profile_all.sh <part>; part 1 = headline + final-scene, part 2 = the million-sphere frame under its two cameras.
# Each is tools/profile_round.sh (kernel trace of the bench command + counter passes); digest afterwards with tools/profile_digest.py.
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT"
if [ "$1" = 1 ]; then
  tools/profile_round.sh r3 || exit 1
  tools/profile_round.sh r3_final-scene --scene final-scene --spp 64 --cpu-spp 2 || exit 1
else
  tools/profile_round.sh r3_c5_across --scene mega-spheres --mega-grid 500 --width 3840 --spp 16 --cpu-spp 1 || exit 1
  tools/profile_round.sh r3_c5_above --scene mega-spheres --mega-grid 500 --width 3840 --spp 16 --cpu-spp 1 --mega-view above || exit 1
fi
# then: tools/profile_digest.py r3; ... r3_final-scene final-scene 64; ... r3_c5_across mega-spheres 16; ... r3_c5_above mega-spheres-above 16
# (the two bench lines of the latter renamed to profiles/r3_c5_{across,above}_bench_line.json) and tools/c5_summary.py on those two lines
