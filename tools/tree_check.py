"""Ad-hoc: ray counts and frame means of one scene through the oracle, the f64 kernels and the f32 kernels (the library's own
tree and, with RTAMD_REFERENCE_TREE=1, the reference's)."""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "rust-raytracing_amd"))
sys.path.insert(0, os.path.join(REPO, "tests"))
import oracle_lib  # noqa: E402
import rtamd  # noqa: E402

name = sys.argv[1]
w = int(sys.argv[2])
spp = int(sys.argv[3])
grid = int(sys.argv[4]) if len(sys.argv) > 4 else 0
hs = rtamd.Scene(name, width=w, spp=spp, mega_grid=grid)
if not os.environ.get("SKIP_ORACLE"):
    os_ = oracle_lib.OracleScene(name, width=w, spp=spp, mega_grid=grid)
    want, cnt, _ = os_.render(0, spp, seed=5)
    print("oracle rays", cnt["rays"], "mean", want.reshape(-1, 3).mean(0) / spp)
ctx = rtamd.Context(0)
ctx.upload(hs)
for prec, nm in ((rtamd.F64, "f64"), (rtamd.F32, "f32")):
    _, got, st = ctx.render(0, spp, precision=prec, seed=5)
    print(nm, "rays", st["rays"], "mean", got.reshape(-1, 3).mean(0) / spp)
