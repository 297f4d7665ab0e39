"""The multi-rank path with the HIP kernels: through the library's communicator entry points (one rank: that is what a one-GPU
box can run over RCCL) and as two ranks sharing GPU 0 in fresh child processes (sums exchanged over gloo: RCCL refuses two
ranks on one device)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_comm_entry_points_single_rank(gpu, rtamd):
    """rt_comm_unique_id / rt_comm_init / rt_reduce_accum / rt_comm_destroy on a communicator of one rank: the reduce must leave
    the colour sums as they are and must not touch the unused fourth value of a pixel's slot."""
    torch = pytest.importorskip("torch")
    sc = rtamd.Scene("cornell-box", width=64, aspect=1.0, spp=4)
    gpu.upload(sc)
    with pytest.raises(rtamd.RtError) as e:
        gpu.reduce_accum(None, 64, 64)  # no communicator yet
    assert e.value.code == 1
    cid = rtamd.comm_unique_id()
    assert len(cid) == 128 and any(cid)
    gpu.comm_init(cid, 1, 0)
    try:
        with pytest.raises(rtamd.RtError):
            gpu.comm_init(cid, 1, 0)  # already in a communicator
        for prec, dt in ((rtamd.F64, torch.float64), (rtamd.F32, torch.float32)):
            accum = torch.zeros(64 * 64 * 4, dtype=dt, device="cuda:0")
            accum.view(-1, 4)[:, 3] = 123.0
            torch.cuda.synchronize()
            gpu.render_accum(accum.data_ptr(), 0, 4, precision=prec, seed=3)
            before = accum.clone()
            gpu.reduce_accum(accum.data_ptr(), 64, 64, precision=prec, root=0)
            assert torch.equal(before, accum) and float(accum.view(-1, 4)[:, :3].sum()) > 0
            with pytest.raises(rtamd.RtError):
                gpu.reduce_accum(accum.data_ptr(), 64, 64, precision=prec, root=1)
    finally:
        gpu.comm_destroy()
    gpu.comm_destroy()  # idempotent


def _bench(args, nproc, tmp):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable]
    if nproc > 1:
        cmd += ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1", "--master-port", "29531"]
    cmd += [os.path.join(REPO, "bench.py"), "--gpus", str(nproc), "--steps", "1", "--warmup", "0", "--no-cpu-baseline"] + args
    out = subprocess.run(cmd, env=env, cwd=tmp, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [ln for ln in out.stdout.splitlines() if ln.startswith('{"metric"')]
    assert len(line) == 1, out.stdout[-2000:]
    return json.loads(line[0])


@pytest.mark.parametrize("name,kw,prec_name", [("cornell-box", dict(width=160, aspect=16 / 9, spp=6), "F64"), ("cornell-box", dict(width=160, aspect=16 / 9, spp=6), "F32"),
                                               ("final-scene", dict(width=120, aspect=4 / 3, spp=3), "F64"), ("random", dict(width=208, spp=4), "F32"),
                                               ("two-spheres", dict(width=96, spp=40), "F64")])
def test_row_bands_tile_the_frame(gpu, rtamd, earth_rgb, name, kw, prec_name):
    """rt_render_params::band_*: three callers share a frame whose height is not a multiple of the band (90 rows in bands of 16).
    Each call fills the rows of its own bands and leaves every other row of the buffer as it was; the three together equal the
    frame rendered in one call up to the order of the additions (pixels keep their whole-frame index in the random streams).
    Covers the flat-walk kernel, the per-lane-walk kernel and the kernel that sums chunks of samples in registers."""
    torch = pytest.importorskip("torch")
    prec = getattr(rtamd, prec_name)
    sc = rtamd.Scene(name, earth=earth_rgb if name == "final-scene" else None, **kw)
    gpu.upload(sc)
    W, H, spp = sc.width, sc.height, sc.spp
    assert H % 16 != 0
    dt = torch.float64 if prec_name == "F64" else torch.float32
    whole = torch.zeros(H * W * 4, dtype=dt, device="cuda:0")
    torch.cuda.synchronize()
    st_whole = gpu.render_accum(whole.data_ptr(), 0, spp, precision=prec, seed=5)
    parts = torch.zeros_like(whole)
    torch.cuda.synchronize()
    rays = cams = 0
    seen = np.zeros(H, dtype=bool)
    for k in (2, 0, 1):
        st = gpu.render_accum(parts.data_ptr(), 0, spp, precision=prec, seed=5, bands=(16, 3, k))
        rows = rtamd.band_rows(H, 16, 3, k)
        assert st["camera_rays"] == len(rows) * W * spp and st["kernel_variant"] == st_whole["kernel_variant"]
        seen[rows] = True
        filled = (parts.view(H, W, 4)[:, :, :3].abs().sum(dim=(1, 2)) > 0).cpu().numpy()
        assert not (filled & ~seen).any(), (k, np.nonzero(filled & ~seen)[0])  # rows of other callers' bands stay untouched
        assert filled[rows].mean() > 0.8  # (a frame may have some black rows of its own)
        rays += st["rays"]
        cams += st["camera_rays"]
    assert seen.all() and cams == W * H * spp
    a, b = whole.view(H, W, 4)[:, :, :3].double().cpu().numpy(), parts.view(H, W, 4)[:, :, :3].double().cpu().numpy()
    if prec_name == "F64":
        assert rays == st_whole["rays"]
        assert np.abs(a - b).max() <= 1e-12 * max(1.0, np.abs(a).max())
    else:  # single precision adds in a different order round differently, nothing more
        assert abs(rays - st_whole["rays"]) <= 1e-6 * st_whole["rays"] + 2
        assert np.abs(a - b).max() <= 2e-5 * max(1.0, np.abs(a).max())
    with pytest.raises(rtamd.RtError):
        gpu.render_accum(parts.data_ptr(), 0, spp, precision=prec, bands=(12, 3, 0))  # not a power of two
    with pytest.raises(rtamd.RtError):
        gpu.render_accum(parts.data_ptr(), 0, spp, precision=prec, bands=(16, 3, 3))  # offset out of range
    with pytest.raises(rtamd.RtError):
        gpu.render_accum(parts.data_ptr(), 0, spp, precision=prec, bands=(16, 3, 0), pipeline=rtamd.PIPELINE_FUSED)


@pytest.mark.parametrize("partition", ["samples", "bands"])
def test_two_ranks_on_one_gpu_equal_one_rank(tmp_path, partition):
    """bench.py's whole step - render the rank's share (a slice of the sample range, or the rows of its bands) with the HIP kernels,
    reduce, resolve - with two ranks (fresh processes, both on GPU 0, gloo) against one rank: the summed radiance agrees to 1e-12
    in f64 (streams are keyed by whole-frame pixel and global sample index; only the order of additions differs), the f32 leg to 1e-5."""
    common = ["--width", "160", "--spp", "8", "--scene", "cornell-box", "--partition", partition, "--band-rows", "8"]
    one = _bench(common + ["--dump-accum", str(tmp_path / "one.npy")], 1, str(tmp_path))
    two = _bench(common + ["--backend", "gloo", "--dump-accum", str(tmp_path / "two.npy")], 2, str(tmp_path))
    assert two["config"]["partition"].startswith(partition)
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2 and two["dtype"] == "f64"
    assert two["config"]["rays_per_step"] == one["config"]["rays_per_step"]
    a, b = np.load(tmp_path / "one.npy"), np.load(tmp_path / "two.npy")
    assert a.shape == (90, 160, 3) and np.abs(a).max() > 0
    assert np.abs(a - b).max() <= 1e-12 * max(1.0, np.abs(a).max())
    assert len(two["per_rank_ms"]["ranks"]) == 2
    e1, e2 = one["f32"]["image_error_vs_f64"], two["f32"]["image_error_vs_f64"]
    assert max(e2["mean_error_rel"]) < 1e-3 and abs(e1["rmse"][0] - e2["rmse"][0]) < 1e-3
