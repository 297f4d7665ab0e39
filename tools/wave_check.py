"""Ad-hoc: does a sample's result depend on which other samples share its wave?"""
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
pixel = np.arange(n, dtype=np.uint32)
sample = np.zeros(n, dtype=np.uint32)
for prec, nm in ((rtamd.F32, "f32"), (rtamd.F64, "f64")):
    full = ctx.sample_radiance(pixel, sample, seed=5, precision=prec)
    rng = np.random.default_rng(1)
    for trial in range(3):
        perm = rng.permutation(n)[: n - 37]
        part = ctx.sample_radiance(pixel[perm], sample[perm], seed=5, precision=prec)
        bad = np.nonzero((part != full[perm]).any(axis=1))[0]
        print(nm, "trial", trial, "samples whose result changed with wave composition:", bad.size)
        for b in bad[:3]:
            print("   pixel", pixel[perm][b], part[b], full[perm][b])
