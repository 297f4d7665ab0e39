"""Grays/s of the three ways to render a per-lane-walk scene: f64 on the caller's tree, f64 on the library's tree, f32."""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "rust-raytracing_amd"))
import rtamd
ctx = rtamd.Context(0)
for scene, kw, spp in (("final-scene", dict(width=800, aspect=1.0), 64), ("random", dict(width=1920), 64)):
    sc = rtamd.Scene(scene, spp=spp, earth_path=os.path.join(REPO, "rust-raytracing_amd", "assets", "earthmap.ppm"), **kw)
    ctx.upload(sc)
    for name, prec in (("f64", rtamd.F64), ("f64_own", rtamd.F64_OWN_TREE), ("f32", rtamd.F32)):
        for rep in range(2):
            st = ctx.render_accum(None, 0, spp, precision=prec)
        print(scene, name, "%.2f Grays/s" % (st["rays"] / st["seconds"] / 1e9), "preset", st["scene_in_lds"], flush=True)
