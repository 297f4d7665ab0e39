"""Ad-hoc: per-(pixel, sample) radiance and ray count, GPU f64 vs the oracle, for a small frame; prints the pairs that differ
and the first path vertex where they part."""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "rust-raytracing_amd"))
sys.path.insert(0, os.path.join(REPO, "tests"))
import oracle_lib  # noqa: E402
import rtamd  # noqa: E402
from scene_pairs import make_pair  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "final-scene"
w = int(sys.argv[2]) if len(sys.argv) > 2 else 100
spp = int(sys.argv[3]) if len(sys.argv) > 3 else 2
seed = int(sys.argv[4]) if len(sys.argv) > 4 else 77
earth = rtamd.load_ppm(os.path.join(REPO, "rust-raytracing_amd/assets/earthmap.ppm"))
hs, os_ = make_pair(rtamd, oracle_lib, name, earth=earth, width=w, spp=spp)
ctx = rtamd.Context(0)
ctx.upload(hs)
n = hs.width * hs.height
pixel = np.tile(np.arange(n, dtype=np.uint32), spp)
sample = np.repeat(np.arange(spp, dtype=np.uint32), n)
got = ctx.sample_radiance(pixel, sample, seed=seed, precision=rtamd.F64)
want = os_.sample_radiance(seed, pixel, sample)
err = np.abs(got[:, :3] - want[:, :3]) / (np.abs(want[:, :3]) + 1e-3)
bad = np.nonzero((err > 1e-9).any(axis=1) | (got[:, 3] != want[:, 3]))[0]
print("pairs", len(pixel), "differing", len(bad), "gpu rays", int(got[:, 3].sum()))
for k in bad[:5]:
    print(" pixel", pixel[k], "sample", sample[k], "gpu", got[k], "oracle", want[k])
    p = ctx.sample_path(pixel[k:k + 1], sample[k:k + 1], seed=seed, precision=rtamd.F64, max_vertices=40)
    g = p[1][0]
    o = os_.sample_path(seed, pixel[k], sample[k], 64)
    for v in range(min(len(o), 40)):
        same = v < len(g) and np.allclose(g[v, :7], o[v, :7], rtol=1e-9, atol=1e-12)
        print("  v%02d oracle t=%.15g obj=%d sub=%d hit=%d | gpu %s %s" % (v, o[v, 6], o[v, 7], o[v, 8], o[v, 9], ("t=%.15g obj=%d sub=%d" % (g[v, 6], g[v, 7], g[v, 8])) if v < len(g) else "-", "" if same else "<-- differs"))
        if not same:
            print("     oracle o,d", o[v, :6]); print("     gpu    o,d", g[v, :6] if v < len(g) else None)
            break
