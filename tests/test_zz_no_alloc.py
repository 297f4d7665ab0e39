"""The allocation-free entry point rt_render_noalloc against rt_render, through the C ABI.

Two calls in a row on the same GPU, the second with a smaller frame and fewer samples: the static scratch (counters, RGB8
staging) keeps the first call's contents, so the second call only comes out right if everything that must start at zero is
cleared.  Bar: ray counts equal; sums within 1e-12 relative in f64 (the same paths; only the order of the atomic additions may
differ) and 1e-4 in f32; RGB8 within 1 level, and equal in all but 1e-4 of the bytes.  The free device memory that
hipMemGetInfo reports must be the same after the call as before it.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _check(rtamd, gpu, torch, scene, spp, prec, seed):
    W, H = scene.width, scene.height
    gpu.upload(scene)
    rgb_ref, acc_ref, st_ref = gpu.render(0, spp, precision=prec, seed=seed)
    dt = torch.float64 if prec == rtamd.F64 else torch.float32
    accum = torch.full((H * W * 4,), 7.0, dtype=dt, device="cuda:0")  # garbage on purpose: the call zeroes it
    pinned = torch.full((H, W, 3), 99, dtype=torch.uint8).pin_memory()  # page-locked: the copy of the image is one DMA, no staging
    rgb = pinned.numpy()
    torch.cuda.synchronize()
    free_before, _ = torch.cuda.mem_get_info(0)  # hipMemGetInfo: device memory free on this GPU
    st = gpu.render_noalloc(accum.data_ptr(), rgb, 0, spp, precision=prec, seed=seed)
    free_after, _ = torch.cuda.mem_get_info(0)
    assert free_after == free_before, "rt_render_noalloc changed the free device memory by %d bytes" % (free_before - free_after)
    got = accum.view(H, W, 4)[:, :, :3].double().cpu().numpy()
    assert st["rays"] == st_ref["rays"] and st["camera_rays"] == W * H * spp
    tol = 1e-12 if prec == rtamd.F64 else 1e-4
    assert np.abs(got - acc_ref).max() <= tol * max(1.0, np.abs(acc_ref).max())
    diff = np.abs(rgb.astype(int) - rgb_ref.astype(int))
    assert diff.max() <= 1 and (diff > 0).mean() <= 1e-4, (diff.max(), (diff > 0).mean())


def test_noalloc_matches_render_twice(gpu, rtamd):
    torch = pytest.importorskip("torch")
    big = rtamd.Scene("cornell-box", width=256, aspect=16.0 / 9.0, spp=8)
    small = rtamd.Scene("cornell-box", width=64, aspect=1.0, spp=3)
    for prec in (rtamd.F64, rtamd.F32):
        _check(rtamd, gpu, torch, big, 8, prec, seed=11)
        _check(rtamd, gpu, torch, small, 3, prec, seed=12)  # smaller frame, fewer samples, right after
    # a per-lane scene as well (stepwise kernel)
    rnd = rtamd.Scene("random", width=96, spp=2)
    _check(rtamd, gpu, torch, rnd, 2, rtamd.F32, seed=5)


def test_noalloc_limits(gpu, rtamd):
    torch = pytest.importorskip("torch")
    sc = rtamd.Scene("cornell-box", width=64, aspect=1.0, spp=1)
    gpu.upload(sc)
    accum = torch.zeros(64 * 64 * 4, dtype=torch.float32, device="cuda:0")
    rgb = np.zeros((64, 64, 3), dtype=np.uint8)
    with pytest.raises(rtamd.RtError) as e:  # beyond the static scratch: 3841 x 2160 pixels
        gpu.render_noalloc(accum.data_ptr(), np.zeros(3841 * 2160 * 3, dtype=np.uint8), 0, 1, width=3841, height=2160)
    assert e.value.code == 1 and "3840*2160" in str(e.value)
    with pytest.raises(rtamd.RtError) as e:  # the queue pipelines need a workspace
        gpu.render_noalloc(accum.data_ptr(), rgb, 0, 1, pipeline=rtamd.PIPELINE_FUSED)
    assert e.value.code == 5
    with pytest.raises(rtamd.RtError):
        gpu.render_noalloc(0, rgb, 0, 1)
