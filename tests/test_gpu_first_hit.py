"""First-hit parity: rt_trace_batch (HIP) against the oracle's BVHNode::hit on the same rays.

Bar: hit flag, object id, nested index, box side, material id and front_face bit-exact; t, p, normal, u, v
within 1e-5 relative for the f32 kernels (north-star tolerance).  The f64 kernels: the hit DISTANCE |dt|*|d| within 1e-11 of
the scale of the numbers involved and p within 1e-11 (observed 1e-15: last-bit differences), t itself within 1e-8 relative - a ray that
starts a thousandth of a unit from a rotated surface has a t whose last six digits are rounding on either side.
In f32 a ray that grazes an edge can legitimately resolve to the neighbouring primitive, so for f32 the exact-id
bar is applied to the rays whose oracle hit is not within 1e-4 (relative) of another candidate; the f64 build
must match on every ray.
"""
import numpy as np
import pytest

from scene_pairs import ALL_SCENES, make_pair, random_rays

pytestmark = pytest.mark.gpu

SCENE_BOXES = {
    "random": ((-14, 0.05, -14), (14, 3, 14)),
    "two-spheres": ((-15, -15, -15), (15, 15, 15)),
    "two-perlin-spheres": ((-15, 0.1, -15), (15, 8, 15)),
    "earth": ((-6, -6, -6), (6, 6, 6)),
    "simple-light": ((-12, 0.1, -12), (28, 12, 12)),
    "cornell-box": ((1, 1, -800), (554, 554, 554)),
    "cornell-smoke": ((1, 1, -800), (554, 554, 554)),
    "final-scene": ((-300, 120, -600), (600, 540, 600)),
}


def compare(gh, oh, rtol, exact_all, label):
    assert np.array_equal(gh["hit"], oh["hit"]) or not exact_all, label
    both = (gh["hit"] == 1) & (oh["hit"] == 1)
    same = np.ones(gh.size, dtype=bool)
    for f in ("hit", "obj", "sub", "side", "material", "front_face"):
        same &= gh[f] == oh[f]
    if exact_all:
        bad = np.nonzero(~same)[0]
        assert bad.size == 0, "%s: %d rays differ in ids, first %s vs %s" % (label, bad.size, gh[bad[:1]], oh[bad[:1]])
    frac = same.mean()
    ok = both & same
    t_err = np.abs(gh["t"][ok] - oh["t"][ok]) / np.maximum(np.abs(oh["t"][ok]), 1e-30)
    assert t_err.size > 0, label
    # p and u,v carry the absolute error of t*|d| and the scene scale; compare them scaled by the scene extent
    scale = np.maximum(np.abs(oh["p"][ok]).max(axis=1), 1.0)
    p_err = np.abs(gh["p"][ok] - oh["p"][ok]).max(axis=1) / scale
    n_err = np.abs(gh["n"][ok] - oh["n"][ok]).max(axis=1)
    return frac, t_err, p_err, n_err, ok


def dist_err(rays, gh, oh, ok):
    """Error of the hit distance |t_gpu - t_oracle| * |d| as a fraction of the scale of the numbers it was computed from: the
    larger of the hit point's coordinates, the origin's and the distance itself."""
    d_len = np.linalg.norm(rays["d"][ok], axis=1)
    scale = np.maximum.reduce([np.abs(oh["p"][ok]).max(axis=1), np.abs(rays["o"][ok]).max(axis=1), np.abs(oh["t"][ok]) * d_len,
                               np.ones(int(ok.sum()))])
    return np.abs(gh["t"][ok] - oh["t"][ok]) * d_len / scale


@pytest.mark.parametrize("name", ALL_SCENES)
def test_first_hit_random_rays_f64(gpu, rtamd, oracle, earth_rgb, name):
    hs, os_ = make_pair(rtamd, oracle, name, earth=earth_rgb)
    gpu.upload(hs)
    rng = np.random.default_rng(1234)
    lo, hi = SCENE_BOXES[name]
    rays = random_rays(rng, 200_000, lo, hi)
    oh = os_.first_hit(rays, seed=99, precision=1)
    gh = gpu.trace_batch(rays, precision=rtamd.F64, seed=99)
    frac, t_err, p_err, n_err, ok = compare(gh, oh, 1e-11, True, name)
    assert (oh["hit"] == 1).mean() > 0.05, "ray batch barely hits the scene"
    assert dist_err(rays, gh, oh, ok).max() < 1e-11 and t_err.max() < 1e-8, (dist_err(rays, gh, oh, ok).max(), t_err.max())
    assert p_err.max() < 1e-11 and n_err.max() < 1e-9
    # (u, v) of a medium hit are whatever an earlier candidate left in the reference's shared record
    # (src/constant_medium.rs never writes them; no texture of an Isotropic material reads them): not compared.
    iso = np.array([int(m["tag"]) == 4 for m in hs.materials()])
    surf = ok & ~iso[np.clip(oh["material"], 0, None)]
    uv_err = np.abs(np.stack([gh["u"], gh["v"]], 1)[surf] - np.stack([oh["u"], oh["v"]], 1)[surf]).max()
    assert uv_err < 1e-9, uv_err


@pytest.mark.parametrize("name", ALL_SCENES)
def test_first_hit_camera_rays_f64(gpu, rtamd, oracle, earth_rgb, name):
    """The rays the sample loop really shoots: oracle-generated primary rays, and the GPU's own primary rays must equal them."""
    hs, os_ = make_pair(rtamd, oracle, name, earth=earth_rgb)
    gpu.upload(hs)
    rng = np.random.default_rng(7)
    n = 100_000
    pixel = rng.integers(0, hs.width * hs.height, size=n).astype(np.uint32)
    sample = rng.integers(0, hs.spp, size=n).astype(np.uint32)
    orays = os_.camera_rays(5, pixel, sample)
    grays = gpu.camera_rays(pixel, sample, seed=5, precision=rtamd.F64)
    assert np.abs(grays["o"] - orays["o"]).max() < 1e-12
    assert np.abs(grays["d"] - orays["d"]).max() < 1e-11
    assert np.abs(grays["time"] - orays["time"]).max() < 1e-15
    oh = os_.first_hit(orays, seed=5, precision=1)
    gh = gpu.trace_batch(orays, precision=rtamd.F64, seed=5)
    frac, t_err, p_err, n_err, ok = compare(gh, oh, 1e-11, True, name)
    assert dist_err(orays, gh, oh, ok).max() < 1e-11 and t_err.max() < 1e-8, (dist_err(orays, gh, oh, ok).max(), t_err.max())


@pytest.mark.parametrize("name", ALL_SCENES)
def test_first_hit_f32_within_tolerance(gpu, rtamd, oracle, earth_rgb, name):
    hs, os_ = make_pair(rtamd, oracle, name, earth=earth_rgb)
    gpu.upload(hs)
    rng = np.random.default_rng(4321)
    n = 100_000
    pixel = rng.integers(0, hs.width * hs.height, size=n).astype(np.uint32)
    sample = rng.integers(0, hs.spp, size=n).astype(np.uint32)
    rays = os_.camera_rays(11, pixel, sample)
    oh = os_.first_hit(rays, seed=11, precision=1)
    gh = gpu.trace_batch(rays, precision=rtamd.F32, seed=11)
    frac, t_err, p_err, n_err, ok = compare(gh, oh, 1e-5, False, name)
    media = name in ("cornell-smoke", "final-scene")  # a medium's sampled t amplifies rounding through ln(U) and the exit distance
    assert frac > (0.995 if not media else 0.98), "%s: only %.4f of f32 rays agree in ids" % (name, frac)
    assert np.quantile(t_err, 0.999) < 1e-5, (name, np.quantile(t_err, 0.999), t_err.max())


@pytest.mark.parametrize("name", ["cornell-box", "cornell-smoke", "simple-light", "random", "final-scene"])
def test_first_hit_f32_as_rendered(gpu, rtamd, oracle, earth_rgb, name):
    """The records the f32 render kernels build (table of planar sides for the Cornell scenes, deferred records elsewhere): rays
    from everywhere inside the scene, so that box sides are met from both sides and under RotateY + Translate."""
    hs, os_ = make_pair(rtamd, oracle, name, earth=earth_rgb)
    gpu.upload(hs)
    rng = np.random.default_rng(2468)
    lo, hi = SCENE_BOXES[name]
    rays = random_rays(rng, 200_000, lo, hi)
    oh = os_.first_hit(rays, seed=3, precision=1)
    gh = gpu.trace_batch(rays, precision=rtamd.F32, seed=3, as_rendered=True)
    frac, t_err, p_err, n_err, ok = compare(gh, oh, 1e-5, False, name)
    # final-scene: its nested trees are dissolved in the f32 image, the sphere cluster's wrappers folded into the sphere centres
    assert frac > (0.995 if name not in ("cornell-smoke", "final-scene") else 0.98), "%s: only %.4f of f32 rays agree in ids and front_face" % (name, frac)
    # origins anywhere, surfaces at any distance: the f32 error of t is that of the hit point (coordinates of a few hundred) over
    # the length of the direction, so it is judged as a distance relative to the scene's extent, like p
    d_len = np.linalg.norm(rays["d"][ok], axis=1)
    scale = np.maximum(np.abs(oh["p"][ok]).max(axis=1), 1.0)
    dist_err = np.abs(gh["t"][ok] - oh["t"][ok]) * d_len / scale
    # `random` stands on a sphere of radius 1000: points near the origin carry the f32 error of that radius
    tol = 1e-5 if name not in ("random", "final-scene") else 1e-4
    assert np.quantile(dist_err, 0.999) < tol, (name, np.quantile(dist_err, 0.999), dist_err.max())
    assert np.quantile(n_err, 0.999) < tol, (name, np.quantile(n_err, 0.999), n_err.max())
    assert np.quantile(p_err, 0.999) < tol, (name, np.quantile(p_err, 0.999), p_err.max())


def test_first_hit_known_answers(gpu, rtamd, oracle):
    """Analytic cases through the full BVH of a two-sphere scene and the Cornell box (both precisions)."""
    hs = rtamd.Scene("cornell-box")
    gpu.upload(hs)
    rays = np.zeros(3, dtype=rtamd.RAY_DTYPE)
    # past both boxes onto the back wall z=555 (object 5, an XY rect)
    rays["o"][0] = (100, 400, -800); rays["d"][0] = (0, 0, 1)
    # up into the light (object 2) at y=554
    rays["o"][1] = (278, 100, 279.5); rays["d"][1] = (0, 2, 0)
    # out of the open front: miss
    rays["o"][2] = (278, 278, 100); rays["d"][2] = (0, 0, -1)
    rays["tmin"] = 0.001; rays["tmax"] = np.inf
    for prec, tol in ((rtamd.F64, 1e-13), (rtamd.F32, 1e-6)):
        h = gpu.trace_batch(rays, precision=prec)
        assert list(h["hit"]) == [1, 1, 0]
        assert h["obj"][0] == 5 and abs(h["t"][0] - 1355.0) < 1355 * tol and h["front_face"][0] == 0  # d . (0,0,1) > 0: back face, normal flipped
        assert np.allclose(h["n"][0], (0, 0, -1))
        assert h["obj"][1] == 2 and abs(h["t"][1] - 227.0) < 227 * tol and h["material"][1] == 3
        assert abs(h["u"][1] - (278 - 213) / 130.0) < 1e-6 and abs(h["v"][1] - (279.5 - 227) / 105.0) < 1e-6


@pytest.mark.parametrize("name,switch", [("random", "RTAMD_NO_CAMERA_ORDER"), ("random", "RTAMD_REFERENCE_TREE"),
                                         ("final-scene", "RTAMD_KEEP_NESTED"), ("cornell-box", "RTAMD_NO_FLAT")])
def test_f32_tree_variants_agree(gpu, rtamd, oracle, earth_rgb, name, switch, monkeypatch):
    """The closest hit does not depend on the tree: the f32 image built with one of the library's upload-time choices switched
    off (child order, own tree, dissolved nested trees, flat leaf lists) reports the same hits as the default image."""
    hs, _ = make_pair(rtamd, oracle, name, earth=earth_rgb)
    rng = np.random.default_rng(97531)
    lo, hi = SCENE_BOXES[name]
    rays = random_rays(rng, 100_000, lo, hi)
    gpu.upload(hs)
    a = gpu.trace_batch(rays, precision=rtamd.F32, seed=5, as_rendered=True)
    monkeypatch.setenv(switch, "1")
    gpu.upload(hs)
    b = gpu.trace_batch(rays, precision=rtamd.F32, seed=5, as_rendered=True)
    monkeypatch.delenv(switch)
    gpu.upload(hs)
    same = (a["hit"] == b["hit"]) & (a["obj"] == b["obj"]) & (a["sub"] == b["sub"]) & (a["side"] == b["side"])
    # exactly coincident surfaces (a box standing on the floor, seen from inside the box): which of the two is reported depends
    # on the order of the tests, i.e. on the tree; such a pair agrees when the distances do
    tie = (a["hit"] == 1) & (b["hit"] == 1) & (np.abs(a["t"] - b["t"]) <= 1e-6 * np.abs(a["t"]))
    assert (same | tie).mean() > 0.999, "%s with %s: %.5f of the rays agree" % (name, switch, (same | tie).mean())
    both = same & (a["hit"] == 1)
    t_err = np.abs(a["t"][both] - b["t"][both]) / np.maximum(np.abs(a["t"][both]), 1e-30)
    assert np.quantile(t_err, 0.999) < 1e-4, (name, switch, np.quantile(t_err, 0.999))

