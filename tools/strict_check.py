"""How far the shipped double-precision small-scene kernels (-ffp-contract=fast, reciprocals and the Lambertian unit vector by
refined hardware estimates) drift from the reference's plain IEEE arithmetic: the same frames rendered by the shipped library and
by a strict build of it (make OUT=$PWD/_build_strict F64_SMALL_FLAGS=-ffp-contract=off HIPFLAGS="... -DRT_STRICT_F64"), each in a
process of its own, compared ray count by ray count and pixel by pixel.
usage: python tools/strict_check.py            (parent: runs both children and compares)
       python tools/strict_check.py --child <out.npz>"""
import os
import subprocess
import sys
import tempfile

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCENES = [("cornell-box", 16), ("cornell-smoke", 16), ("two-spheres", 16), ("two-perlin-spheres", 16), ("earth", 16), ("simple-light", 16)]


def child(out):
    sys.path.insert(0, os.path.join(REPO, "rust-raytracing_amd"))
    import rtamd
    ctx = rtamd.Context(0)
    res = {}
    for name, spp in SCENES:
        sc = rtamd.Scene(name, width=960, aspect=16.0 / 9.0, spp=spp, earth_path=os.path.join(REPO, "rust-raytracing_amd/assets/earthmap.ppm"))
        ctx.upload(sc)
        _, acc, st = ctx.render(0, spp, precision=rtamd.F64, seed=77, want_rgb=False)
        res[name + "_acc"] = acc
        res[name + "_rays"] = np.array([st["rays"], st["scene_in_lds"]], dtype=np.int64)
    np.savez(out, **res)


def main():
    if len(sys.argv) > 2 and sys.argv[1] == "--child":
        return child(sys.argv[2])
    tmp = tempfile.mkdtemp()
    outs = {}
    for tag, build in (("shipped", "_build"), ("strict", "_build_strict")):
        env = dict(os.environ, RTAMD_BUILD_DIR=os.path.join(REPO, "rust-raytracing_amd", build))
        out = os.path.join(tmp, tag + ".npz")
        subprocess.check_call([sys.executable, os.path.abspath(__file__), "--child", out], env=env, timeout=600)
        outs[tag] = np.load(out)
    print("scene                 preset  rays(shipped)   rays rel diff  pixels differing >1e-9  max rel pixel diff  channel-mean rel diff")
    for name, spp in SCENES:
        a, b = outs["shipped"][name + "_acc"], outs["strict"][name + "_acc"]
        ra, rb = outs["shipped"][name + "_rays"], outs["strict"][name + "_rays"]
        rel = np.abs(a - b) / (np.abs(b) + 1e-3)
        ma, mb = a.reshape(-1, 3).mean(0), b.reshape(-1, 3).mean(0)
        print("%-22s %5d %14d %14.3g %22.3g %19.3g %22.3g" % (name, ra[1], ra[0], abs(int(ra[0]) - int(rb[0])) / rb[0], (rel > 1e-9).any(axis=2).mean(), rel.max(),
                                                         (np.abs(ma - mb) / mb).max()))


if __name__ == "__main__":
    main()
