"""Exactly coincident surfaces in double precision.

The default double-precision image of a large scene walks the library's own four-wide tree, not the caller's: the closest hit does
not depend on the tree - unless two different leaves report exactly the same distance, where the reference gives the hit to the
leaf its left-then-right walk tests last, provided its boxes let the walk get there (src/bvh.rs:108-119, src/aabb.rs:38-45: a
box is passed over when t_max <= t_min).  Such rays are flagged by the four-wide walk and traced again by the list walk of the
caller's tree in the reference's order.  Checked here:
  * as-rendered first hits of final-scene (whose ground blocks share faces and edges) against the oracle: every id equal;
  * rays constructed to hit, in exact arithmetic, the common edge of two ground blocks of equal height (a four-way tie among two
    top faces and two side faces): the default image and the list walk (RTAMD_F64_LIST) report the same object, side and record;
  * a render in which two overlapping coplanar rectangles of different materials face the camera, so that thousands of camera
    rays tie: the rays are traced again by the list walk (rt_stats.rays_retraced > 0) and the frame equals the list walk's frame.
"""
import numpy as np
import pytest

from custom_scene import EditedScene
from scene_pairs import make_pair, random_rays
from test_gpu_first_hit import ID_FIELDS, SCENE_BOXES, compare, dist_err

pytestmark = pytest.mark.gpu


def test_as_rendered_first_hits_of_final_scene_match_the_oracle(gpu, rtamd, oracle, earth_rgb):
    hs, os_ = make_pair(rtamd, oracle, "final-scene", earth=earth_rgb)
    gpu.upload(hs)
    assert gpu.render_accum(None, 0, 1, precision=rtamd.F64, seed=3)["scene_in_lds"] == 12
    rng = np.random.default_rng(1234)
    lo, hi = SCENE_BOXES["final-scene"]
    rays = random_rays(rng, 200_000, lo, hi)
    # rays that start inside the ground blocks as well: there shared faces are reachable
    low = random_rays(rng, 100_000, (-900, 0.5, -900), (900, 60, 900))
    for label, batch in (("above", rays), ("among the blocks", low)):
        oh = os_.first_hit(batch, seed=99, precision=1)
        gh = gpu.trace_batch(batch, precision=rtamd.F64, seed=99, as_rendered=True)
        frac, t_err, p_err, n_err, ok = compare(gh, oh, 1e-11, True, "final-scene as rendered, " + label)
        assert dist_err(batch, gh, oh, ok).max() < 1e-11 and p_err.max() < 1e-11 and n_err.max() < 1e-9


def _flatten_two_blocks(rtamd, hs, height):
    """Two ground blocks that are neighbours along x, both cut down to the same height (their node boxes still contain them)."""
    ed = EditedScene(rtamd, hs)
    boxes = [i for i in range(len(ed.shapes)) if int(ed.shapes["tag"][i]) == 5]
    by_corner = {(float(ed.shapes["p"][i][0]), float(ed.shapes["p"][i][2])): i for i in boxes}
    for (x0, z0), a in sorted(by_corner.items()):
        b = by_corner.get((x0 + 100.0, z0))
        if b is not None and min(ed.shapes["p"][a][4], ed.shapes["p"][b][4]) > height and -400 <= x0 <= 200 and -300 <= z0 <= 300:
            ed.shapes["p"][a][4] = height
            ed.shapes["p"][b][4] = height
            return ed, a, b
    raise AssertionError("no pair of neighbouring blocks taller than %g" % height)


def test_constructed_ties_resolve_like_the_list_walk(gpu, rtamd, oracle, earth_rgb, monkeypatch):
    hs, _ = make_pair(rtamd, oracle, "final-scene", earth=earth_rgb)
    h = 8.25
    ed, a, b = _flatten_two_blocks(rtamd, hs, h)
    x_edge = float(ed.shapes["p"][b][0])  # = a's x1: a multiple of 100
    z0 = float(ed.shapes["p"][a][2])
    n = 4096
    rng = np.random.default_rng(5)
    rays = np.zeros(n, dtype=rtamd.RAY_DTYPE)
    # origin and direction in binary fractions so that o + t d lands on the edge (x_edge, h, z) exactly at t = 128
    dx = rng.choice([-8, -5, -3, -2, -1, 1, 2, 3, 5, 8], size=n) / 16.0  # never 0: a ray inside a face's plane is another matter
    dy = -rng.integers(1, 9, size=n) / 32.0
    dz = rng.integers(-4, 5, size=n) / 64.0
    z = z0 + rng.integers(8, 92, size=n).astype(np.float64)
    rays["d"] = np.stack([dx, dy, dz], 1)
    rays["o"] = np.stack([x_edge - 128.0 * dx, h - 128.0 * dy, z - 128.0 * dz], 1)
    rays["tmin"], rays["tmax"] = 0.001, np.inf
    assert np.array_equal(rays["o"][:, 0] + 128.0 * rays["d"][:, 0], np.full(n, x_edge)) and np.array_equal(rays["o"][:, 1] + 128.0 * rays["d"][:, 1], np.full(n, h))
    gpu.upload(ed)
    got = gpu.trace_batch(rays, precision=rtamd.F64, seed=1, as_rendered=True)
    monkeypatch.setenv("RTAMD_F64_LIST", "1")
    gpu.upload(ed)
    want = gpu.trace_batch(rays, precision=rtamd.F64, seed=1, as_rendered=True)
    monkeypatch.delenv("RTAMD_F64_LIST")
    gpu.upload(hs)
    on_pair = np.isin(want["sub"], [int(ed.objects["sub_id"][k]) for k in range(len(ed.objects)) if int(ed.objects["geom"][k]) in (a, b) and not (int(ed.objects["flags"][k]) & 8)])
    assert (want["hit"] == 1).all() and on_pair.mean() > 0.9 and (np.abs(want["t"] - 128.0) < 1e-9)[on_pair].all(), on_pair.mean()
    assert len(set(zip(want["sub"][on_pair], want["side"][on_pair]))) >= 2  # the ties do not all go to one face
    for f in ID_FIELDS + ("t", "p", "n"):
        assert np.array_equal(got[f], want[f]), f


def test_render_with_coplanar_overlapping_rectangles(gpu, rtamd, oracle, earth_rgb, monkeypatch):
    """The glass ball of final-scene and the fog inside it (two objects with the same box in the caller's tree) are replaced by two
    overlapping rectangles in the plane z = 145, facing the camera, one emitting: every ray into the overlap meets both at exactly
    the same distance."""
    hs, _ = make_pair(rtamd, oracle, "final-scene", earth=earth_rgb, width=200, aspect=1.0, spp=4)
    ed = EditedScene(rtamd, hs)
    objs, shapes = ed.objects, ed.shapes
    ball = [k for k in range(len(objs)) if not (int(objs["flags"][k]) & 8) and int(shapes["tag"][objs["geom"][k]]) == 0
            and tuple(shapes["p"][objs["geom"][k]][:4]) == (360.0, 150.0, 145.0, 70.0)]
    assert len(ball) == 2, ball  # the dielectric sphere and the medium bounded by the same sphere
    light = int(np.nonzero(ed.materials["tag"] == 3)[0][0])
    ground = int(shapes["material"][[i for i in range(len(shapes)) if int(shapes["tag"][i]) == 5][0]])
    new = np.zeros(2, dtype=shapes.dtype)
    new["tag"] = 2  # XY rectangles: x0 x1 y0 y1 k
    new["p"][0][:5] = (300.0, 400.0, 90.0, 210.0, 145.0)
    new["p"][1][:5] = (330.0, 425.0, 100.0, 200.0, 145.0)
    new["material"] = (ground, light)
    ed.shapes = np.concatenate([shapes, new])
    for k, g in zip(ball, (len(shapes), len(shapes) + 1)):
        ed.objects["geom"][k] = g
        ed.objects["flags"][k] = 0
    gpu.upload(ed)
    _, a, sa = gpu.render(0, 4, precision=rtamd.F64, seed=8)
    assert sa["scene_in_lds"] == 12 and sa["rays_retraced"] > 1000, sa
    monkeypatch.setenv("RTAMD_F64_LIST", "1")
    gpu.upload(ed)
    _, b, sb = gpu.render(0, 4, precision=rtamd.F64, seed=8)
    monkeypatch.delenv("RTAMD_F64_LIST")
    gpu.upload(hs)
    assert sb["scene_in_lds"] in (7, 8) and sb["rays_retraced"] == 0
    assert sa["rays"] == sb["rays"], (sa["rays"], sb["rays"])
    assert (np.abs(a - b) / (np.abs(b) + 1e-3)).max() < 1e-12
    # and the overlap shows ONE of the two materials (which one is the reference's rule), not a mixture of them from pixel to pixel
    assert np.abs(a).max() > 0
