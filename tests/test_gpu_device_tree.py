"""The four-wide tree built on the GPU at upload (gpu_build.hip; scenes of 100 000 plain objects and more, RTAMD_GPU_BUILD_MIN for
another threshold) and the sorted search for exact copies (RTAMD_SORT_MIN): same closest hits as the host builder's tree and as
the oracle.  The million-sphere tests of test_gpu_bench_configs.py run on the device-built tree as well (the default there)."""
import numpy as np
import pytest

from scene_pairs import make_pair, random_rays

pytestmark = pytest.mark.gpu
ID_FIELDS = ("hit", "obj", "sub", "side", "material", "front_face")


@pytest.mark.parametrize("grid,view", [(12, 0), (40, 0), (90, 1)])
def test_device_built_tree_finds_the_same_hits(gpu, rtamd, oracle, monkeypatch, grid, view):
    """Sphere fields of 577, 6 401 and 32 401 objects: first hits of camera rays and of rays in all directions on the tree built
    on the device against the oracle (ids identical in double precision, single precision outside silhouettes) and against the
    same image on the host builder's tree (double precision: bit for bit; the closest hit does not depend on the tree); a small
    frame in both precisions; the statistics say which builder made the tree."""
    kw = dict(mega_grid=grid, mega_view=view, width=240, spp=2)
    hs, os_ = make_pair(rtamd, oracle, "mega-spheres", **kw)
    rng = np.random.default_rng(grid)
    n = 60_000
    pixel = rng.integers(0, hs.width * hs.height, size=n).astype(np.uint32)
    sample = rng.integers(0, 64, size=n).astype(np.uint32)
    cam = os_.camera_rays(4, pixel, sample)
    anyway = random_rays(rng, n, (-grid, 0.05, -grid), (grid, 2.5, grid))
    results = {}
    for builder in ("device", "host"):
        if builder == "device":
            monkeypatch.setenv("RTAMD_GPU_BUILD_MIN", "64")
        else:
            monkeypatch.delenv("RTAMD_GPU_BUILD_MIN")
            monkeypatch.setenv("RTAMD_NO_GPU_BUILD", "1")
        gpu.upload(hs)
        _, acc64, st64 = gpu.render(0, 2, precision=rtamd.F64, seed=6)
        _, acc32, st32 = gpu.render(0, 2, precision=rtamd.F32, seed=6)
        for st in (st64, st32):
            assert bool(st["kernel_variant"] & rtamd.VARIANT_DEVICE_TREE) == (builder == "device"), (builder, st)
        assert st64["scene_in_lds"] == 11 and st32["scene_in_lds"] == 9
        results[builder] = [gpu.trace_batch(r, precision=p, seed=4, as_rendered=True) for r in (cam, anyway) for p in (rtamd.F64, rtamd.F32)] + [acc64, acc32, st64, st32]
    monkeypatch.delenv("RTAMD_NO_GPU_BUILD")
    dev, host = results["device"], results["host"]
    for k, rays in ((0, cam), (2, anyway)):
        oh = os_.first_hit(rays, seed=4, precision=1)
        for f in ID_FIELDS:  # double precision: the oracle's answer in every id, and the host tree's in every bit
            assert np.array_equal(dev[k][f], oh[f]), (grid, k, f, int((dev[k][f] != oh[f]).sum()))
        assert dev[k].tobytes() == host[k].tobytes()
        same32 = np.ones(n, dtype=bool)
        for f in ID_FIELDS:
            same32 &= dev[k + 1][f] == oh[f]
        assert same32.mean() > 0.999, same32.mean()
    assert dev[6]["rays"] == host[6]["rays"]
    assert (np.abs(dev[4] - host[4]) / (np.abs(host[4]) + 1e-3)).max() < 1e-12
    assert abs(int(dev[7]["rays"]) - int(host[7]["rays"])) <= 2e-3 * host[7]["rays"]


def test_device_built_tree_is_the_same_on_every_upload(gpu, rtamd, monkeypatch):
    """Child nodes are numbered by prefix sums, not by the order in which threads arrive: two uploads give the same tree, so the same
    box and leaf counts for every ray."""
    monkeypatch.setenv("RTAMD_GPU_BUILD_MIN", "64")
    hs = rtamd.Scene("mega-spheres", mega_grid=50, width=160, spp=1)
    rays = random_rays(np.random.default_rng(1), 20_000, (-50, 0.05, -50), (50, 2.5, 50))
    visits = []
    for _ in range(2):
        gpu.upload(hs)
        h = gpu.trace_batch(rays, precision=rtamd.F32, seed=1, as_rendered=True, visits=True)
        visits.append((h["u"].copy(), h["v"].copy()))
    assert np.array_equal(visits[0][0], visits[1][0]) and np.array_equal(visits[0][1], visits[1][1])
    assert visits[0][0].mean() > 4  # boxes were tested


def test_sorted_search_for_copies(gpu, rtamd, oracle, monkeypatch):
    """The `random` scene holds its three large spheres 22 times each (src/scenes.rs:197-217).  With the search for exact copies run
    through the device's sort (RTAMD_SORT_MIN=1; the default for 65 536 objects and more) the images are the ones the hash-table
    search gives: same presets, same hits, bit for bit."""
    hs, os_ = make_pair(rtamd, oracle, "random")
    rays = random_rays(np.random.default_rng(2), 50_000, (-12, 0.05, -12), (12, 3, 12))
    gpu.upload(hs)
    a64 = gpu.trace_batch(rays, precision=rtamd.F64, seed=2, as_rendered=True, visits=True)
    a32 = gpu.trace_batch(rays, precision=rtamd.F32, seed=2, as_rendered=True, visits=True)
    monkeypatch.setenv("RTAMD_SORT_MIN", "1")
    gpu.upload(hs)
    b64 = gpu.trace_batch(rays, precision=rtamd.F64, seed=2, as_rendered=True, visits=True)
    b32 = gpu.trace_batch(rays, precision=rtamd.F32, seed=2, as_rendered=True, visits=True)
    monkeypatch.delenv("RTAMD_SORT_MIN")
    gpu.upload(hs)
    assert a64.tobytes() == b64.tobytes() and a32.tobytes() == b32.tobytes()  # visit counts included: the same leaves were left out
    oh = os_.first_hit(rays, seed=2, precision=1)
    for f in ID_FIELDS:
        assert np.array_equal(b64[f], oh[f]), f
