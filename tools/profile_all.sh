#!/bin/bash
# The round's profiles in one go: tools/profile_all.sh <part>; part 1 = headline + earth, part 2 = random + final-scene + 1M spheres.
# Each is tools/profile_round.sh (kernel trace of the bench command + counter passes); digest afterwards with tools/profile_digest.py.
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT"
if [ "$1" = 1 ]; then
  tools/profile_round.sh r2 || exit 1
  tools/profile_round.sh r2_earth --scene earth --spp 256 --cpu-spp 8 || exit 1
else
  tools/profile_round.sh r2_random --scene random --spp 64 --cpu-spp 4 || exit 1
  tools/profile_round.sh r2_final-scene --scene final-scene --spp 64 --cpu-spp 2 || exit 1
  tools/profile_round.sh r2_c5 --scene mega-spheres --mega-grid 500 --width 3840 --spp 16 --cpu-spp 1 || exit 1
fi
