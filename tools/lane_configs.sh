# bench.py on the per-lane-walk configurations (random, final-scene, the million-sphere field), one JSON line each into
# gpurun_out/lane_configs.jsonl, then the table of tools/configs_table.py.
set -e
cd "$(dirname "$0")/.."; mkdir -p gpurun_out; OUT=gpurun_out/lane_configs.jsonl; : > $OUT
run() { timeout -k 10 400 python3 bench.py --steps 2 --warmup 1 "$@" 2>>gpurun_out/lane_configs.err | grep '^{"metric"' >> "$OUT"; }
run --scene random --spp 64 --no-cpu-baseline
run --scene final-scene --spp 64 --no-cpu-baseline
run --scene mega-spheres --mega-grid 500 --width 3840 --spp 16 --no-cpu-baseline
run --scene random --width 400 --spp 100 --no-cpu-baseline
run --scene final-scene --width 800 --aspect 1.0 --spp 64 --no-cpu-baseline
python3 tools/configs_table.py $OUT
