"""Builds the same scene twice: by the product's host library and by the oracle's own restatement of scenes.rs."""
import numpy as np

ALL_SCENES = ["random", "two-spheres", "two-perlin-spheres", "earth", "simple-light", "cornell-box", "cornell-smoke", "final-scene"]


def make_pair(rtamd, oracle, name, earth=None, **kw):
    need_earth = name in ("earth", "final-scene")
    e = earth if need_earth else None
    hs = rtamd.Scene(name, earth=e, **kw)
    os_ = oracle.OracleScene(name, earth=e, want_f32=kw.pop("want_f32", False), **kw)
    assert (hs.width, hs.height, hs.spp, hs.depth) == (os_.width, os_.height, os_.spp, os_.depth)
    return hs, os_


def random_rays(rng, n, box_lo, box_hi, pixel_space=1 << 20):
    """Rays with origins in a box and un-normalised directions of varied length (the reference never normalises)."""
    rays = np.zeros(n, dtype=[("o", "<f8", 3), ("d", "<f8", 3), ("time", "<f8"), ("tmin", "<f8"), ("tmax", "<f8"),
                              ("pixel", "<u4"), ("sample", "<u4"), ("bounce", "<u4"), ("pad", "<u4")])
    rays["o"] = rng.uniform(box_lo, box_hi, size=(n, 3))
    d = rng.normal(size=(n, 3))
    d *= rng.uniform(0.05, 12.0, size=(n, 1)) / np.linalg.norm(d, axis=1, keepdims=True)
    rays["d"] = d
    rays["time"] = rng.uniform(0, 1, size=n)
    rays["tmin"] = 0.001
    rays["tmax"] = np.inf
    rays["pixel"] = rng.integers(0, pixel_space, size=n)
    rays["sample"] = rng.integers(0, 1024, size=n)
    rays["bounce"] = rng.integers(0, 50, size=n)
    return rays
