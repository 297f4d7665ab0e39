"""Ad-hoc: f32 frame sums vs f64 frame sums for one scene under both pipelines (same streams)."""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "rust-raytracing_amd"))
import rtamd  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "cornell-box"
w = int(sys.argv[2]) if len(sys.argv) > 2 else 120
spp = int(sys.argv[3]) if len(sys.argv) > 3 else 8
ctx = rtamd.Context(0)
sc = rtamd.Scene(name, width=w, spp=spp, earth_path=os.path.join(REPO, "rust-raytracing_amd/assets/earthmap.ppm"))
ctx.upload(sc)
_, ref, sref = ctx.render(0, spp, precision=rtamd.F64, pipeline=rtamd.PIPELINE_FUSED, seed=5)
for pl, nm in ((rtamd.PIPELINE_FUSED, "fused"), (rtamd.PIPELINE_SORTED, "sorted")):
    for rep in range(2):
        _, a, st = ctx.render(0, spp, precision=rtamd.F32, pipeline=pl, seed=5)
        print(name, nm, "rep", rep, "rays f32 %d f64 %d" % (st["rays"], sref["rays"]), "mean f32", (a / spp).mean(), "f64", (ref / spp).mean(),
              "preset", st["scene_in_lds"])
