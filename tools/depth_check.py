"""Ad-hoc: ray counts of the queue kernels vs the one-thread-per-path kernel, by depth limit."""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "rust-raytracing_amd"))
import rtamd  # noqa: E402

ctx = rtamd.Context(0)
sc = rtamd.Scene("cornell-box", width=64, spp=2)
ctx.upload(sc)
n = sc.width * sc.height
pixel = np.tile(np.arange(n, dtype=np.uint32), 2)
sample = np.repeat(np.arange(2, dtype=np.uint32), n)
for prec, nm in ((rtamd.F32, "f32"), (rtamd.F64, "f64")):
    for depth in (1, 2, 3, 4, 6):
        _, acc, st = ctx.render(0, 2, precision=prec, depth=depth, seed=5)
        rad = ctx.sample_radiance(pixel, sample, seed=5, precision=prec, depth=depth)
        ref = np.zeros((n, 3))
        np.add.at(ref, pixel, rad[:, :3])
        print(nm, "depth", depth, "queue rays", st["rays"], "path rays", int(rad[:, 3].sum()), "max accum diff", np.abs(acc.reshape(-1, 3) - ref).max())
