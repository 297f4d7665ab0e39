"""Ad-hoc: per-sample comparison of the f32 and f64 kernels (same streams) to locate single-precision drift."""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "rust-raytracing_amd"))
import rtamd  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "cornell-box"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 400000
ctx = rtamd.Context(0)
sc = rtamd.Scene(name, earth_path=os.path.join(REPO, "rust-raytracing_amd/assets/earthmap.ppm"))
ctx.upload(sc)
rng = np.random.default_rng(0)
pixel = rng.integers(0, sc.width * sc.height, size=n).astype(np.uint32)
sample = rng.integers(0, 64, size=n).astype(np.uint32)
a = ctx.sample_radiance(pixel, sample, seed=3, precision=rtamd.F32)
b = ctx.sample_radiance(pixel, sample, seed=3, precision=rtamd.F64)
dl = a[:, 3] - b[:, 3]
print(name, "mean rays f32 %.4f f64 %.4f" % (a[:, 3].mean(), b[:, 3].mean()))
print(" paths with different length: %.4f%%  (longer in f32 %.4f%%, shorter %.4f%%)" % (100 * (dl != 0).mean(), 100 * (dl > 0).mean(), 100 * (dl < 0).mean()))
rel = np.abs(a[:, :3] - b[:, :3]).max(1) / (np.abs(b[:, :3]).max(1) + 1e-3)
print(" radiance differs >1e-3: %.4f%%; mean f32 %s f64 %s" % (100 * (rel > 1e-3).mean(), a[:, :3].mean(0), b[:, :3].mean(0)))
v, c = np.unique(np.clip(dl, -5, 5), return_counts=True)
print(" length difference (clipped to +-5):", dict(zip(v.astype(int).tolist(), c.tolist())))
worst = np.argsort(-np.abs(dl))[:6]
for k in worst:
    print("  pixel %d sample %d: rays f32 %d f64 %d" % (pixel[k], sample[k], a[k, 3], b[k, 3]))
k = worst[0]
np.set_printoptions(precision=6, suppress=True, linewidth=200)
for prec, nm in ((rtamd.F32, "f32"), (rtamd.F64, "f64")):
    rad, verts = ctx.sample_path(pixel[k:k + 1], sample[k:k + 1], seed=3, precision=prec, max_vertices=5)
    print(nm, rad[0])
    print(verts[0])
