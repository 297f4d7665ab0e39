#!/bin/bash
# bench.py on one configuration with different numbers of cursor grabs per wave (RTAMD_GRABS_PER_WAVE); one line per setting.
# usage: tools/grab_sweep.sh "<bench args>" g1 g2 ...   (run through gpurun; every step has its own time limit, the first failure ends the sweep)
set -e
ARGS=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT"
for g in "$@"; do
  line=$(RTAMD_GRABS_PER_WAVE=$g timeout -k 10 200 python3 bench.py $ARGS --no-cpu-baseline | grep '^{"metric"')
  echo "grabs_per_wave=$g $(echo "$line" | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print("%.0f Mrays/s  %.3f ms/step  kernel %.3f ms" % (d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"]))')"
done
