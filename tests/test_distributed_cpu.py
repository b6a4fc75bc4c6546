"""CPU tier: the N>1 path (sharding + the all-gather collect) with world_size-2 and world_size-8 gloo processes."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from gps_optimize_slam_amd import distributed as D

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_ranges_partition():
    for total in (0, 1, 7, 8, 1000, 10_000_000):
        for world in (1, 2, 3, 8):
            r = [D.shard_range(total, k, world) for k in range(world)]
            assert r[0][0] == 0 and r[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(r, r[1:]))
            sizes = [b - a for a, b in r]
            assert max(sizes) - min(sizes) <= 1


def _fake_fused(traj_ids, n):
    """stand-in for the kernel output: depends only on the global trajectory id -> shard invariant by construction"""
    t = torch.as_tensor(traj_ids, dtype=torch.float64)
    pos = t[:, None, None] * 1e3 + torch.arange(n, dtype=torch.float64)[None, :, None] + torch.tensor([0.1, 0.2, 0.3], dtype=torch.float64)
    quat = torch.sin(t)[:, None, None] + torch.arange(n, dtype=torch.float64)[None, :, None] * 1e-3 + torch.zeros(4, dtype=torch.float64)
    return pos.contiguous(), quat.contiguous()


def _worker(rank, world, port, total, n, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    r, w, _ = D.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    lo, hi = D.shard_range(total, rank, world)
    pos, quat = _fake_fused(np.arange(lo, hi), n)
    pa, qa = D.allgather_poses(pos, quat)
    full_p, full_q = _fake_fused(np.arange(total), n)
    assert torch.equal(pa.reshape(total, n, 3), full_p) and torch.equal(qa.reshape(total, n, 4), full_q)
    # chunked form with a checksum sink
    sums = []
    D.allgather_poses(pos, quat, chunk_trajs=3, sink=lambda k, p, q: sums.append(float(p.sum() + q.sum())))
    assert abs(sum(sums) - float(full_p.sum() + full_q.sum())) < 1e-6 * abs(float(full_p.sum()))
    # time-major tensors (N, C, B): the trajectory axis is the LAST one (traj_dim=-1)
    pt, qt = pos.permute(1, 2, 0).contiguous(), quat.permute(1, 2, 0).contiguous()
    pa, qa = D.allgather_poses(pt, qt, traj_dim=-1)
    assert torch.equal(torch.cat(list(pa), dim=2), full_p.permute(1, 2, 0)) and torch.equal(torch.cat(list(qa), dim=2), full_q.permute(1, 2, 0))
    got = []
    D.allgather_poses(pt, qt, chunk_trajs=2, traj_dim=-1, sink=lambda k, p, q: got.append((p.shape, q.shape)))
    assert got[0] == ((world, n, 3, 2), (world, n, 4, 2)) and len(got) == -(-(hi - lo) // 2)
    with pytest.raises(ValueError):
        D.allgather_poses(pt, qt, traj_dim=5)
    flat = torch.empty((world * pos.numel(),), dtype=torch.float64)
    D.all_gather_flat(flat, pos.reshape(-1))
    assert torch.equal(flat.view(total, n, 3), full_p)
    assert int(D.all_reduce(torch.tensor([rank + 1])).item()) == world * (world + 1) // 2
    t = D.max_over_ranks(1.0 + rank, "cpu")
    assert t == float(world)
    D.barrier()
    open(os.path.join(out_dir, f"ok{rank}"), "w").write("ok")
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize("world,total,n", [(2, 10, 5), (8, 32, 3)])
def test_world2_gloo_allgather(tmp_path, world, total, n):
    """world 2: the contract's minimum; world 8: the rank count of the driver's scaling run (one process per GPU of a node), rehearsed on
    the host cores -- the shard ranges, the all-gather collect in both layouts, its chunked form and the max-over-ranks timing."""
    port = 29000 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(world, port, total, n, str(tmp_path)), nprocs=world, join=True)
    assert all((tmp_path / f"ok{r}").exists() for r in range(world))
