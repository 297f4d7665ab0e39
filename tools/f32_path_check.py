"""Ad-hoc: where do f32 paths part from f64 ones?  Prints the first vertices of a few samples whose path lengths differ."""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "rust-raytracing_amd"))
import rtamd  # noqa: E402

name = sys.argv[1]
w = int(sys.argv[2])
grid = int(sys.argv[3]) if len(sys.argv) > 3 else 0
hs = rtamd.Scene(name, width=w, spp=1, mega_grid=grid)
ctx = rtamd.Context(0)
ctx.upload(hs)
n = hs.width * hs.height
pixel = np.arange(n, dtype=np.uint32)
sample = np.zeros(n, dtype=np.uint32)
a = ctx.sample_radiance(pixel, sample, seed=5, precision=rtamd.F32)
b = ctx.sample_radiance(pixel, sample, seed=5, precision=rtamd.F64)
bad = np.nonzero(a[:, 3] != b[:, 3])[0]
print("paths", n, "different length", len(bad), "f32 rays", a[:, 3].sum(), "f64 rays", b[:, 3].sum())
for k in bad[:4]:
    pa = ctx.sample_path(pixel[k:k + 1], sample[k:k + 1], seed=5, precision=rtamd.F32, max_vertices=12)[1][0]
    pb = ctx.sample_path(pixel[k:k + 1], sample[k:k + 1], seed=5, precision=rtamd.F64, max_vertices=12)[1][0]
    print("pixel", pixel[k], "f32 len", a[k, 3], "f64 len", b[k, 3])
    for v in range(6):
        print("  v%d f32 t=%.6g obj=%d mat=%d hit=%d o=(%.4f %.4f %.4f) | f64 t=%.6g obj=%d mat=%d hit=%d" % (v, pa[v, 6], pa[v, 7], pa[v, 9], pa[v, 11], pa[v, 0], pa[v, 1], pa[v, 2], pb[v, 6], pb[v, 7], pb[v, 9], pb[v, 11]))
