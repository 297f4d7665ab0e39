"""Algorithmic flops per ray of every bench scene: event counts of the CPU oracle on samples 0..1 of the bench frame times the
fixed per-event weights of bench.py.  Writes profiles/flops_per_ray.json (read by bench.py when the CPU leg is skipped)."""
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
sys.path.insert(0, os.path.join(REPO, "rust-raytracing_amd"))
import bench  # noqa: E402
import oracle_lib  # noqa: E402
import rtamd  # noqa: E402

earth = rtamd.load_ppm(os.path.join(REPO, "rust-raytracing_amd", "assets", "earthmap.ppm"))
out = {}
for scene in ("cornell-box", "cornell-smoke", "simple-light", "two-perlin-spheres", "earth", "random", "final-scene"):
    osc = oracle_lib.OracleScene(scene, width=1920, aspect=16.0 / 9.0, spp=2, depth=50, earth=earth if scene in ("earth", "final-scene") else None)
    _, cnt, secs = osc.render(0, 2, seed=0, threads=7)
    out[scene] = bench.flops_per_ray(cnt)
    print(scene, out[scene], "rays", cnt["rays"], "%.1fs" % secs, flush=True)
os.makedirs(os.path.join(REPO, "profiles"), exist_ok=True)
with open(os.path.join(REPO, "profiles", "flops_per_ray.json"), "w") as f:
    json.dump(out, f, indent=1, sort_keys=True)
