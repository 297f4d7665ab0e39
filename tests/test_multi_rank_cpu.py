"""The N > 1 path on CPU: two gloo ranks shard the sample range exactly as bench.py does, reduce their accumulation
buffers onto rank 0 and resolve there.  The oracle stands in for the renderer (no GPU here); what is under test is the
sharding arithmetic, the reduce, and that the result does not depend on the number of ranks (the random streams are keyed
by the global sample index)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))


def _worker(rank, world, port, spp, out_path, partition="samples"):
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), "rust-raytracing_amd"))
    import oracle_lib
    import rtamd  # host-side arithmetic only: nothing here needs the HIP library
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sc = oracle_lib.OracleScene("cornell-box", width=40, spp=spp)
    if partition == "samples":
        s0, s1 = rank * spp // world, (rank + 1) * spp // world      # bench.py's split
        acc, cnt, _ = sc.render(s0, s1, seed=11, threads=2)
    else:  # bands of 4 rows dealt out in turn: a rank keeps the rows of its own bands (the oracle renders whole frames)
        acc, cnt, _ = sc.render(0, spp, seed=11, threads=2)
        mine = np.zeros(acc.shape[0], dtype=bool)
        mine[rtamd.band_rows(acc.shape[0], 4, world, rank)] = True
        acc[~mine] = 0.0
        cnt["rays"] = cnt["rays"] if rank == 0 else 0
    t = torch.from_numpy(acc)
    rays = torch.tensor([cnt["rays"]], dtype=torch.float64)
    dist.reduce(t, dst=0, op=dist.ReduceOp.SUM)
    dist.all_reduce(rays, op=dist.ReduceOp.SUM)
    if rank == 0:
        np.save(out_path, np.concatenate([t.numpy().ravel(), rays.numpy()]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sample_sharding_reduces_to_single_rank_result(oracle, tmp_path, world):
    spp = 7  # not divisible by the world size on purpose
    out = str(tmp_path / "acc.npy")
    mp.spawn(_worker, args=(world, 29500 + world, spp, out), nprocs=world, join=True)
    got = np.load(out)
    sc = oracle.OracleScene("cornell-box", width=40, spp=spp)
    want, cnt, _ = sc.render(0, spp, seed=11, threads=2)
    assert got[-1] == cnt["rays"]
    assert np.abs(got[:-1].reshape(want.shape) - want).max() <= 1e-12 * max(1.0, want.max())
    # shares cover the range exactly once
    parts = [(r * spp // world, (r + 1) * spp // world) for r in range(world)]
    assert parts[0][0] == 0 and parts[-1][1] == spp and all(parts[i][1] == parts[i + 1][0] for i in range(world - 1))


@pytest.mark.parametrize("world", [2, 3])
def test_row_bands_reduce_to_single_rank_result_bit_for_bit(oracle, tmp_path, world):
    """Rows shared out in bands (rt_render_params::band_*) and summed by the same reduce: every pixel is rendered by one rank and
    the others add zeros, so the result equals the one-rank frame in every bit; the bands cover each row exactly once."""
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), "rust-raytracing_amd"))
    import rtamd
    spp = 3
    out = str(tmp_path / "acc.npy")
    mp.spawn(_worker, args=(world, 29510 + world, spp, out, "bands"), nprocs=world, join=True)
    got = np.load(out)
    sc = oracle.OracleScene("cornell-box", width=40, spp=spp)
    want, cnt, _ = sc.render(0, spp, seed=11, threads=2)
    assert got[-1] == cnt["rays"]
    assert np.array_equal(got[:-1].reshape(want.shape), want)
    for height, rows, stride in ((40, 4, world), (90, 16, 3), (1080, 16, 8), (7, 8, 2), (5, 1, 4)):
        owner = np.full(height, -1)
        for k in range(stride):
            r = rtamd.band_rows(height, rows, stride, k)
            assert (owner[r] == -1).all()
            owner[r] = k
        assert (owner >= 0).all() and (owner == (np.arange(height) // rows) % stride).all()
