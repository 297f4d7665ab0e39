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


def test_two_ranks_on_one_gpu_equal_one_rank(tmp_path):
    """bench.py's whole step - render the rank's sample range with the HIP kernels, reduce, resolve - with two ranks (fresh
    processes, both on GPU 0, gloo) against one rank: the summed radiance agrees to 1e-12 in f64 (sample streams are keyed by
    the global sample index; only the order of additions differs) and the f32 leg to 1e-5."""
    common = ["--width", "160", "--spp", "8", "--scene", "cornell-box"]
    one = _bench(common + ["--dump-accum", str(tmp_path / "one.npy")], 1, str(tmp_path))
    two = _bench(common + ["--backend", "gloo", "--dump-accum", str(tmp_path / "two.npy")], 2, str(tmp_path))
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2 and two["dtype"] == "f64"
    assert two["config"]["rays_per_step"] == one["config"]["rays_per_step"]
    a, b = np.load(tmp_path / "one.npy"), np.load(tmp_path / "two.npy")
    assert a.shape == (90, 160, 3) and np.abs(a).max() > 0
    assert np.abs(a - b).max() <= 1e-12 * max(1.0, np.abs(a).max())
    assert len(two["per_rank_ms"]["ranks"]) == 2
    e1, e2 = one["f32"]["image_error_vs_f64"], two["f32"]["image_error_vs_f64"]
    assert max(e2["mean_error_rel"]) < 1e-3 and abs(e1["rmse"][0] - e2["rmse"][0]) < 1e-3
