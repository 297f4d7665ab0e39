"""Ad-hoc GPU check: render a scene at modest size in both precisions and print throughput + means."""
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "rust-raytracing_amd"))
import rtamd  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "cornell-box"
width = int(sys.argv[2]) if len(sys.argv) > 2 else 600
spp = int(sys.argv[3]) if len(sys.argv) > 3 else 64
aspect = float(sys.argv[4]) if len(sys.argv) > 4 else 0.0
ctx = rtamd.Context(0)
sc = rtamd.Scene(name, width=width, spp=spp, aspect=aspect, earth_path=os.path.join(REPO, "rust-raytracing_amd/assets/earthmap.ppm"))
ctx.upload(sc)
for prec, pname in ((rtamd.F32, "f32"), (rtamd.F64, "f64")):
    for pipe, plname in ((rtamd.PIPELINE_FUSED, "fused"), (rtamd.PIPELINE_SORTED, "sorted")):
        for rep in range(2):
            t0 = time.time()
            rgb, acc, st = ctx.render(0, spp, precision=prec, pipeline=pipe)
            dt = time.time() - t0
        m = (acc / spp).reshape(-1, 3).mean(0)
        print("%s %s %s %dx%d spp=%d: %.3fs wall, %.1f Mrays/s (rays %d, %.2f/sample), bounce_ms %.1f launches %d lds %d, mean %s"
              % (name, pname, plname, sc.width, sc.height, spp, dt, st["rays"] / st["seconds"] / 1e6, st["rays"],
                 st["rays"] / st["camera_rays"], st["bounce_kernel_ms"], st["bounce_kernel_launches"], st["lds_bytes"], m), flush=True)
    rtamd.save_image(os.path.join(REPO, "gpurun_out", "%s_%s.png" % (name, pname)), rgb)
