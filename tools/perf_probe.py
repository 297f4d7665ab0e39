"""Ad-hoc: render the headline frame at a given sample count and print the library's own timing split."""
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "rust-raytracing_amd"))
import rtamd  # noqa: E402

scene = sys.argv[1] if len(sys.argv) > 1 else "cornell-box"
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 128
prec = {"f64": rtamd.F64, "f64own": rtamd.F64_OWN_TREE}.get(sys.argv[3] if len(sys.argv) > 3 else "f32", rtamd.F32)
spass = int(sys.argv[4]) if len(sys.argv) > 4 else 0
width = int(sys.argv[5]) if len(sys.argv) > 5 else 1920
pipe = int(os.environ.get("PROBE_PIPELINE", "0"))
ctx = rtamd.Context(0)
sc = rtamd.Scene(scene, width=width, aspect=16.0 / 9.0, spp=spp, earth_path=os.path.join(REPO, "rust-raytracing_amd/assets/earthmap.ppm"),
                 mega_grid=int(os.environ.get("PROBE_MEGA", "0")), mega_view=int(os.environ.get("PROBE_MEGA_VIEW", "0")))
t_up = time.time()
ctx.upload(sc)
print("scene: %d objects, %d nodes, upload %.2f s" % (len(sc.objects()), len(sc.nodes()), time.time() - t_up), flush=True)
for rep in range(int(os.environ.get("PROBE_REPS", "3"))):
    t0 = time.time()
    st = ctx.render_accum(None, 0, spp, precision=prec, samples_per_pass=spass, pipeline=pipe)
    dt = time.time() - t0
print("pipe %d " % pipe + "%s %s spp=%d pass=%d: wall %.1f ms, lib %.1f ms, bounce kernels %.1f ms (%d launches), gen %.1f ms, passes %d, rays %.3fG, %d rays traced again at ties -> %.2f Grays/s (kernel-only %.2f)"
      % (scene, {rtamd.F32: "f32", rtamd.F64: "f64"}.get(prec, "f64 own tree"), spp, spass, dt * 1e3, st["seconds"] * 1e3, st["bounce_kernel_ms"], st["bounce_kernel_launches"], st["gen_kernel_ms"],
         st["passes"], st["rays"] / 1e9, st["rays_retraced"], st["rays"] / st["seconds"] / 1e9, st["rays"] / (st["bounce_kernel_ms"] * 1e-3) / 1e9))
