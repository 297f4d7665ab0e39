"""Times scene construction and rt_upload_scene for the million-sphere scene (RTAMD_TIMING=1 prints the stages of the upload)."""
import os, sys, time
os.environ["RTAMD_TIMING"] = "1"
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "rust-raytracing_amd"))
import rtamd
grid = int(sys.argv[1]) if len(sys.argv) > 1 else 500
t = time.perf_counter()
s = rtamd.Scene("mega-spheres", width=3840, spp=16, mega_grid=grid)
print("scene construction (reference tree included): %.3f s, %d objects" % (time.perf_counter() - t, len(s.objects())), flush=True)
ctx = rtamd.Context(0)
for k in range(2):
    t = time.perf_counter()
    ctx.upload(s)
    print("rt_upload_scene #%d: %.3f s" % (k, time.perf_counter() - t), flush=True)
ctx.close()
