"""The kernels and frame sizes the benchmark times, held to the oracle.

bench.py's frames are large: 1920x1080 at 1024 samples (2.1 G paths) and 3840x2160 for the million-sphere field.  At that size
the launcher picks the kernel instantiations that hand out paths in constant grabs of 2048 per wave (RT_VARIANT_FIXED_GRAB), which
no frame the oracle can follow would ever select, and pixel indices run to 8.3 M.  Here
  * RTAMD_FIXED_GRAB=1 forces those instantiations on small frames, compared with the oracle like every other frame: identical
    ray counts, per-pixel sums within 1e-9 relative (double precision), single precision by the usual tolerance;
  * primary rays and whole sample paths of random (pixel, sample) pairs of the 1920x1080 and 3840x2160 frames are compared with
    the oracle's (origin 1e-12, direction 1e-11, radiance 1e-9): a sample costs the same whatever the frame's size;
  * the million-sphere field is checked at its full size: first hits of 100 000 camera rays of the 3840x2160 frame against the
    oracle's own tree of the same million spheres (ids exact; single precision outside ties), and one whole 3840x2160 frame.
"""
import numpy as np
import pytest

from scene_pairs import make_pair
from test_gpu_first_hit import ID_FIELDS, compare, dist_err

pytestmark = pytest.mark.gpu

CHAOTIC = {"final-scene": 1e-4, "random": 1e-4, "mega-spheres": 1e-3}

# scene -> (width, spp, kwargs, variant bits the double-precision launch must report besides FIXED_GRAB)
FIXED = {
    "cornell-box": (120, 8, {}, 0),            # k_paths_persistent<double, boxes, true, false>: the headline's kernel
    "cornell-smoke": (120, 8, {}, 0),
    "simple-light": (200, 8, {}, 0),           # spheres preset, unchunked
    "two-perlin-spheres": (200, 8, {}, 0),
    "earth": (200, 8, {}, 2),                  # chunked
    "two-spheres": (200, 8, {}, 2),
    "random": (160, 4, {}, 4),                 # stepwise, four-wide walk
    "final-scene": (100, 2, {}, 4),
    "mega-spheres": (160, 4, {"mega_grid": 20}, 4),
}


@pytest.mark.parametrize("name", list(FIXED))
def test_fixed_grab_kernels_match_the_oracle(gpu, rtamd, oracle, earth_rgb, name, monkeypatch):
    w, spp, kw, bits = FIXED[name]
    hs, os_ = make_pair(rtamd, oracle, name, earth=earth_rgb, width=w, spp=spp, **kw)
    gpu.upload(hs)
    want, cnt, _ = os_.render(0, spp, seed=77)
    _, plain, st0 = gpu.render(0, spp, precision=rtamd.F64, seed=77)
    assert not (st0["kernel_variant"] & rtamd.VARIANT_FIXED_GRAB)  # a frame this small never selects them by itself
    monkeypatch.setenv("RTAMD_FIXED_GRAB", "1")
    rgb, got, st = gpu.render(0, spp, precision=rtamd.F64, seed=77)
    _, got32, st32 = gpu.render(0, spp, precision=rtamd.F32, seed=77)
    own = gpu.render(0, spp, precision=rtamd.F64_OWN_TREE, seed=77) if st["scene_in_lds"] >= 3 else None
    monkeypatch.delenv("RTAMD_FIXED_GRAB")
    assert st["kernel_variant"] & ~rtamd.VARIANT_NARROW_BOXES == (rtamd.VARIANT_FIXED_GRAB | bits), st["kernel_variant"]  # (the box-test flag says nothing about the instantiation)
    assert st32["kernel_variant"] & rtamd.VARIANT_FIXED_GRAB
    allow = CHAOTIC.get(name, 0.0)
    assert abs(int(st["rays"]) - int(cnt["rays"])) <= allow * cnt["rays"], (st["rays"], cnt["rays"])
    assert st["camera_rays"] == cnt["camera_rays"] == hs.width * hs.height * spp
    err = np.abs(got - want) / (np.abs(want) + 1e-3)
    assert (err.reshape(-1, 3).max(axis=1) > 1e-9).mean() <= allow, err.max()
    off = np.abs(rgb.astype(int) - oracle.resolve(want, spp).astype(int)).reshape(-1, 3).max(axis=1)
    assert (off > 1).mean() <= allow and (off > 0).mean() <= 1e-4 + allow
    # the same paths as the variable-grab kernel of the same scene: only the order of the atomic additions may differ
    assert st["rays"] == st0["rays"]
    assert (np.abs(got - plain) / (np.abs(plain) + 1e-3)).max() < 1e-12
    # single precision, by tolerance
    a, b = np.clip(got32 / spp, 0, 1), np.clip(want / spp, 0, 1)
    mean_err = np.abs(a.reshape(-1, 3).mean(0) - b.reshape(-1, 3).mean(0)) / b.reshape(-1, 3).mean(0)
    assert mean_err.max() < (1e-2 if name in CHAOTIC else 2e-3), (name, mean_err)
    assert np.sqrt(((a - b) ** 2).mean()) < 0.02
    assert abs(int(st32["rays"]) - int(cnt["rays"])) <= 5e-3 * cnt["rays"]
    if own is not None:  # double precision on the library's own tree: the other instantiation bench.py times for these scenes
        _, acc_own, st_own = own
        assert st_own["kernel_variant"] & rtamd.VARIANT_FIXED_GRAB
        assert abs(int(st_own["rays"]) - int(cnt["rays"])) <= 2e-3 * cnt["rays"]
        mo, mw = np.clip(acc_own / spp, 0, 1).reshape(-1, 3).mean(0), b.reshape(-1, 3).mean(0)
        assert (np.abs(mo - mw) / mw).max() < 1e-2, (mo, mw)


@pytest.mark.parametrize("name,kw", [("cornell-box", {}), ("mega-spheres", {"mega_grid": 30}), ("final-scene", {}), ("earth", {})])
@pytest.mark.parametrize("width", [1920, 3840])
def test_primary_rays_and_paths_at_bench_frame_sizes(gpu, rtamd, oracle, earth_rgb, name, kw, width):
    """Pixel indices of the benchmark's frames (up to 8 294 399): the row / column split of a pixel index, the jitter divisors
    W - 1 and H - 1 (src/render.rs:232-233) and Camera::get_ray (src/camera.rs:62-72) against the oracle's, then whole paths."""
    hs, os_ = make_pair(rtamd, oracle, name, earth=earth_rgb, width=width, aspect=16.0 / 9.0, spp=1024, **kw)
    assert (hs.width, hs.height) == (width, width * 9 // 16)
    gpu.upload(hs)
    rng = np.random.default_rng(width)
    n = 100_000
    pixel = rng.integers(0, hs.width * hs.height, size=n).astype(np.uint32)
    pixel[:4] = [0, hs.width - 1, hs.width * (hs.height - 1), hs.width * hs.height - 1]  # the four corners
    pixel[4:8] = [hs.width, 2 * hs.width - 1, (1 << 23) % (hs.width * hs.height), hs.width * hs.height - hs.width]
    sample = rng.integers(0, 1024, size=n).astype(np.uint32)
    orays = os_.camera_rays(5, pixel, sample)
    grays = gpu.camera_rays(pixel, sample, seed=5, precision=rtamd.F64)
    assert np.abs(grays["o"] - orays["o"]).max() < 1e-12
    assert np.abs(grays["d"] - orays["d"]).max() < 1e-11
    assert np.abs(grays["time"] - orays["time"]).max() < 1e-15
    g32 = gpu.camera_rays(pixel, sample, seed=5, precision=rtamd.F32)
    # single precision: a direction is a sum of camera vectors with coordinates of up to several hundred; half a dozen roundings of those
    scale = max(1.0, np.abs(hs.camera()[:12]).max())
    # (a camera with a lens: a disk candidate within rounding of the rim is accepted in one precision and not in the other, one ray
    # in 10^5 or so; those rays are real rays of the same camera, just of another candidate)
    d_err, o_err = np.abs(g32["d"] - orays["d"]).max(axis=1), np.abs(g32["o"] - orays["o"]).max(axis=1)
    assert np.quantile(d_err, 0.9999) < 4e-7 * scale and np.quantile(o_err, 0.9999) < 1e-7 * scale, (d_err.max(), o_err.max())
    assert (d_err > 4e-7 * scale).sum() <= (0 if hs.camera()[21] == 0.0 else 10)
    m = 20_000
    want = os_.sample_radiance(21, pixel[:m], sample[:m])
    got = gpu.sample_radiance(pixel[:m], sample[:m], seed=21, precision=rtamd.F64)
    err = np.abs(got[:, :3] - want[:, :3]) / (np.abs(want[:, :3]) + 1e-3)
    bad = (err > 1e-9).any(axis=1) | (got[:, 3] != want[:, 3])
    assert bad.sum() <= CHAOTIC.get(name, 0.0) * m, (name, width, int(bad.sum()), err.max())


def test_frame_sums_at_1080p_rows(gpu, rtamd, oracle):
    """One sample of every pixel of the headline frame itself (cornell-box 1920x1080) through the path kernel, against the oracle's
    frame of the same sample: 2 073 600 paths, identical ray counts, sums within 1e-9."""
    hs, os_ = make_pair(rtamd, oracle, "cornell-box", width=1920, aspect=16.0 / 9.0, spp=1024)
    gpu.upload(hs)
    want, cnt, _ = os_.render(1000, 1001, seed=3)
    _, got, st = gpu.render(1000, 1001, precision=rtamd.F64, seed=3)
    assert st["rays"] == cnt["rays"] and st["camera_rays"] == 1920 * 1080
    assert (np.abs(got - want) / (np.abs(want) + 1e-3)).max() < 1e-9


@pytest.mark.parametrize("view", [0, 1])
def test_million_spheres_at_3840x2160_against_the_oracle(gpu, rtamd, oracle, view):
    """The synthetic large scene as configured: one million spheres (mega_grid = 500), 3840x2160, under both of its cameras (0: across
    the field, 1: straight down on all of it, every sphere smaller than a pixel).  Oracle and host library build
    the same million spheres and the same reference tree independently.  First hits of 100 000 camera rays of the 4K frame:
    double precision (the library's four-wide tree, the reference's leaf arithmetic) identical to the oracle in every id, single
    precision identical outside ties; then one whole 4K frame in each precision."""
    hs, os_ = make_pair(rtamd, oracle, "mega-spheres", width=3840, aspect=16.0 / 9.0, spp=1, mega_grid=500, mega_view=view)
    assert (hs.width, hs.height) == (3840, 2160) and os_.n_objects == 1_000_001
    gpu.upload(hs)
    rng = np.random.default_rng(500)
    n = 100_000
    pixel = rng.integers(0, hs.width * hs.height, size=n).astype(np.uint32)
    sample = rng.integers(0, 1024, size=n).astype(np.uint32)
    rays = os_.camera_rays(9, pixel, sample)
    grays = gpu.camera_rays(pixel, sample, seed=9, precision=rtamd.F64)
    assert np.abs(grays["o"] - rays["o"]).max() < 1e-12 and np.abs(grays["d"] - rays["d"]).max() < 1e-11
    # secondary rays as well: from points above the field in all directions
    more = rays.copy()
    more["o"] = rng.uniform((-450, 0.3, -450), (450, 1.5, 450), size=(n, 3))
    d = rng.normal(size=(n, 3))
    more["d"] = d / np.linalg.norm(d, axis=1, keepdims=True) * rng.uniform(0.05, 4.0, size=(n, 1))
    for label, batch in (("camera", rays), ("secondary", more)):
        oh = os_.first_hit(batch, seed=9, precision=1)
        assert (oh["hit"] == 1).mean() > 0.4, label
        for prec, tag in ((rtamd.F64, "default"), (rtamd.F64_OWN_TREE, "RT_F64_OWN_TREE")):
            gh = gpu.trace_batch(batch, precision=prec, seed=9, as_rendered=True)
            frac, t_err, p_err, n_err, ok = compare(gh, oh, 1e-11, False, label)
            same = np.ones(n, dtype=bool)
            for f in ID_FIELDS:
                same &= gh[f] == oh[f]
            both = (gh["hit"] == 1) & (oh["hit"] == 1)
            tie = np.zeros(n, dtype=bool)
            tie[both] = dist_err(batch, gh, oh, both) <= 1e-12
            print("\n1M spheres, %s rays, f64 on the %s: %d of %d differ in ids (%d not ties)" % (label, tag, int((~same).sum()), n, int((~(same | tie)).sum())))
            assert (same | tie).all() and (~same).mean() < 1e-4
            assert dist_err(batch, gh, oh, ok).max() < 1e-11 and p_err.max() < 1e-11 and n_err.max() < 1e-9
        g32 = gpu.trace_batch(batch, precision=rtamd.F32, seed=9, as_rendered=True)
        same = np.ones(n, dtype=bool)
        for f in ID_FIELDS:
            same &= g32[f] == oh[f]
        both = same & (oh["hit"] == 1)
        t_rel = np.abs(g32["t"][both] - oh["t"][both]) / np.abs(oh["t"][both])
        print("1M spheres, %s rays, f32: %.4f%% differ in ids; t rel err max %.3g, 99.9%% %.3g" % (label, 100 * (1 - same.mean()), t_rel.max(), np.quantile(t_rel, 0.999)))
        assert same.mean() > 0.999  # silhouettes of a million small spheres: a grazing ray may resolve either way in single precision
        if label == "camera":
            assert np.quantile(t_rel, 0.999) < 1e-5
        # origins anywhere make t itself ill-conditioned (a ray that starts next to a sphere): judged as a distance, like the other
        # single-precision first-hit tests
        assert dist_err(batch, g32, oh, both).max() < 1e-5
    # one whole frame of the configured size
    _, a64, s64 = gpu.render(0, 1, precision=rtamd.F64, seed=13)
    _, a32, s32 = gpu.render(0, 1, precision=rtamd.F32, seed=13)
    _, aown, sown = gpu.render(0, 1, precision=rtamd.F64_OWN_TREE, seed=13)
    assert s64["camera_rays"] == s32["camera_rays"] == sown["camera_rays"] == 3840 * 2160
    assert s64["scene_in_lds"] == 11 and s32["scene_in_lds"] == 9
    assert abs(int(s32["rays"]) - int(s64["rays"])) <= 5e-3 * s64["rays"], (s32["rays"], s64["rays"])
    assert abs(int(sown["rays"]) - int(s64["rays"])) <= 1e-3 * s64["rays"], (sown["rays"], s64["rays"])
    m64 = np.clip(a64, 0, 1).reshape(-1, 3).mean(0)
    for other in (a32, aown):
        mo = np.clip(other, 0, 1).reshape(-1, 3).mean(0)
        assert (np.abs(mo - m64) / m64).max() < 5e-3, (mo, m64)
    # rows of the 4K frame against the oracle's render of the same rows (every path of 16 rows: 61 440 paths)
    j0, j1 = 1072, 1088  # rows counted from the bottom, as j of src/render.rs:225; image row y = H - 1 - j
    want, cnt, _ = os_.render(0, 1, seed=13, rows=(j0, j1))
    band = a64[hs.height - j1:hs.height - j0]
    wband = want[hs.height - j1:hs.height - j0]
    assert np.abs(wband).max() > 0
    err = np.abs(band - wband) / (np.abs(wband) + 1e-3)
    assert (err.reshape(-1, 3).max(axis=1) > 1e-9).mean() <= 1e-3, err.max()
