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


def _worker(rank, world, port, spp, out_path):
    sys.path.insert(0, HERE)
    import oracle_lib
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sc = oracle_lib.OracleScene("cornell-box", width=40, spp=spp)
    s0, s1 = rank * spp // world, (rank + 1) * spp // world      # bench.py's split
    acc, cnt, _ = sc.render(s0, s1, seed=11, threads=2)
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
